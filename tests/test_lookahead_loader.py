"""The look-ahead protocol that gives the reference's unmodified train_one_epoch the no-read-back path:
pcdet.datasets.LookaheadLoader hands every batch its successor, Detector3DTemplate.forward enqueues the successor's sparse prologue
(tools/train_utils/train_utils.py:26-44 is the caller this serves). CPU: the loader's contract; GPU: a SECOND training loop through
model_fn_decorator() gives the same losses with and without it, and every step after the first finds its batch prefetched."""
import numpy as np
import pytest
import torch


def test_lookahead_loader_yields_the_same_batches_and_names_the_successor():
    from pcdet.datasets import LookaheadLoader, SyntheticDataset, build_synthetic_dataloader
    ds = SyntheticDataset(num_frames=5, n_points=300)
    plain = build_synthetic_dataloader(ds, 2)
    look = LookaheadLoader(build_synthetic_dataloader(ds, 2))
    assert len(look) == len(plain) == 3 and look.dataset is ds and look.batch_size == 2 and look.num_workers == 0
    a, b = list(plain), list(look)
    assert len(a) == len(b) == 3
    for i, (x, y) in enumerate(zip(a, b)):
        assert list(x['frame_id']) == list(y['frame_id']) and np.array_equal(x['points'], y['points'])
        assert y['_crb_next'] is (b[i + 1] if i + 1 < len(b) else None)           # the very dict the next iteration yields
    assert list(LookaheadLoader(build_synthetic_dataloader(SyntheticDataset(num_frames=0, n_points=10), 2))) == []
    # a second epoch over the same object starts over
    assert [list(y['frame_id']) for y in look] == [list(x['frame_id']) for x in a]


@pytest.mark.gpu
def test_training_loop_through_the_lookahead_loader_prefetches_and_gives_the_same_losses(dev):
    from pcdet.datasets import LookaheadLoader, SyntheticDataset, build_synthetic_dataloader
    from pcdet.model_cfgs import second_cfg
    from pcdet.models import build_network, model_fn_decorator
    ds = SyntheticDataset(num_frames=8, n_points=6000)
    losses, hits = {}, {}
    was = torch.are_deterministic_algorithms_enabled()
    torch.use_deterministic_algorithms(True, warn_only=True)       # (bit-reproducible training steps: tests/test_determinism.py)
    try:
        for look in (False, True):
            torch.manual_seed(0)
            model = build_network(second_cfg().MODEL, 3, ds).to(dev).train()
            opt = torch.optim.SGD(model.parameters(), lr=1e-3)
            loader = build_synthetic_dataloader(ds, 2)
            if look:
                loader = LookaheadLoader(loader)
            fn = model_fn_decorator()
            out = []
            it = iter(loader)                                   # (the reference's loop: next(dataloader_iter), model_func(model, batch))
            for _ in range(len(loader)):
                batch = next(it)
                opt.zero_grad()
                loss, tb, _ = fn(model, batch)
                loss.backward()
                opt.step()
                out.append(float(loss.detach()))
            losses[look], hits[look] = out, model.__dict__.get('_crb_prefetch_hits', 0)
    finally:
        torch.use_deterministic_algorithms(was)
    assert hits[False] == 0 and hits[True] == len(losses[True]) - 1, hits
    # the same kernels on the same inputs in another order of launches, in the deterministic mode (the default mode's two atomically
    # summed vendor weight gradients make the fourth loss of two runs differ by up to 1e-4 once a ReLU flips): equal, bit for bit
    assert losses[True] == losses[False], (losses[True], losses[False])
