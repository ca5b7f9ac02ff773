"""GPU: PartA2FCHead of the mirror (HIP RoI-aware pooling, subm rulebook + gather-GEMM on the B*N RoI grids, dense scatter)
against ref_partA2.npz, written by the reference's own PartA2FCHead (tests/golden/make_goldens.py:gen_partA2, compiled entry
points answered by the oracle). Weights are regenerated from the generator's seeded stream (same names, same shapes).
Tolerances: pooled features 1e-6 (avg = sum / count in f32); head outputs 2e-4 relative to the largest magnitude (two
27-offset convs + three FC layers in f32 against the oracle's double accumulation)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _close(got, ref, rel):
    np.testing.assert_allclose(got, ref, rtol=rel, atol=rel * float(np.abs(ref).max()))


def _head_and_batch(dev):
    from golden._constants import parta2_cfg, parta2_inputs, seeded_state
    from pcdet.models.roi_heads import PartA2FCHead
    head = PartA2FCHead(input_channels=32, model_cfg=parta2_cfg(), num_class=1)
    g = np.load(os.path.join(GOLD, 'ref_partA2.npz'))
    assert sorted(head.state_dict().keys()) == list(g['pa2_keys'])          # the reference's parameter names
    head.load_state_dict(seeded_state(head, 43))
    head.to(dev)
    inp = parta2_inputs()

    def batch():
        return {k: (torch.from_numpy(v.copy()).to(dev) if isinstance(v, np.ndarray) else v) for k, v in inp.items()}
    return head, batch, g


def test_partA2_head_eval_matches_reference_golden(dev):
    head, batch, g = _head_and_batch(dev)
    head.eval()
    with torch.no_grad():
        part, rpn = head.roiaware_pool(batch())
        bd = head(batch())
    np.testing.assert_allclose(part.cpu().numpy(), g['pa2_pooled_part'], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(rpn.cpu().numpy(), g['pa2_pooled_rpn'])                  # max pooling: exact
    assert int((g['pa2_pooled_part'].sum(-1) != 0).sum()) > 200
    _close(bd['batch_cls_preds'].cpu().numpy(), g['pa2_eval_cls'], 2e-4)
    _close(bd['batch_box_preds'].cpu().numpy(), g['pa2_eval_box'], 2e-4)
    assert bd['cls_preds_normalized'] is False


def test_partA2_head_train_mode_matches_reference_golden_and_backpropagates(dev):
    """batch-statistics BatchNorm over the occupied cells of ALL RoI grids; then a backward pass reaches the point
    features through the max pooling, the convs and the FC stack (finite, non-zero)"""
    head, batch, g = _head_and_batch(dev)
    head.train()
    head.assign_targets = lambda bdict: {'rois': bdict['rois'], 'roi_labels': bdict['roi_labels']}
    b = batch()
    b['point_features'].requires_grad_(True)
    head(b)
    fr = head.forward_ret_dict
    _close(fr['rcnn_cls'].detach().cpu().numpy(), g['pa2_train_cls'], 5e-4)
    _close(fr['rcnn_reg'].detach().cpu().numpy(), g['pa2_train_reg'], 5e-4)
    (fr['rcnn_cls'].square().sum() + fr['rcnn_reg'].square().sum()).backward()
    gp = b['point_features'].grad
    assert torch.isfinite(gp).all() and float(gp.abs().sum()) > 0
    for n, p in head.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
