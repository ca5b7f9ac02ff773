#!/usr/bin/env python
"""VERDICT r03 item 2 (peak memory of the PV-RCNN step): bytes allocated at the end of every module of the detector's module
list in the forward of one training step at bs=16, the peak inside each module, and the ten largest tensors the autograd graph
holds at the end of the forward (saved for backward).  usage: python tools/mem_pvrcnn.py [--batch 16]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

GB = 2.0 ** 30


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--points', type=int, default=20000)
    a = ap.parse_args()
    from pcdet.datasets import SyntheticDataset
    from pcdet.datasets.synthetic import kitti_batch
    from pcdet.model_cfgs import pv_rcnn_cfg
    from pcdet.models import build_network
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = build_network(pv_rcnn_cfg().MODEL, 3, SyntheticDataset(num_frames=2)).to(dev)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
    pts, off, gt = kitti_batch(100, a.batch, a.points)
    bidx = np.repeat(np.arange(a.batch, dtype=np.float32), np.diff(off))
    batch = {'points': torch.from_numpy(np.concatenate([bidx[:, None], pts], 1)).to(dev),
             'point_frame_offsets': torch.from_numpy(off).to(dev), 'gt_boxes': torch.from_numpy(gt).to(dev),
             'batch_size': a.batch, 'point_frame_counts_host': np.diff(off).tolist()}

    def step(hooks):
        opt.zero_grad(set_to_none=True)
        ret, tb, _ = model(dict(batch))
        if hooks:
            torch.cuda.synchronize()
            print('end of forward: allocated %.2f GB, peak so far %.2f GB' % (torch.cuda.memory_allocated() / GB, torch.cuda.max_memory_allocated() / GB))
            torch.cuda.reset_peak_memory_stats()
        ret['loss'].backward()
        if hooks:
            torch.cuda.synchronize()
            print('end of backward: allocated %.2f GB, peak inside backward %.2f GB' % (torch.cuda.memory_allocated() / GB, torch.cuda.max_memory_allocated() / GB))
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()

    for _ in range(2):
        step(False)
    torch.cuda.synchronize()
    print('resident before the step (weights, AdamW state, inputs): %.2f GB' % (torch.cuda.memory_allocated() / GB))
    state = {}

    def pre(name):
        def f(mod, inp):
            torch.cuda.synchronize()
            state[name] = torch.cuda.memory_allocated()
            torch.cuda.reset_peak_memory_stats()
        return f

    def post(name):
        def f(mod, inp, out):
            torch.cuda.synchronize()
            now = torch.cuda.memory_allocated()
            print('%-28s kept +%.2f GB (allocated %.2f GB), peak inside %.2f GB' % (name, (now - state[name]) / GB, now / GB, torch.cuda.max_memory_allocated() / GB))
        return f
    hs = []
    mods = list(model.module_list)
    for m in mods:
        n = type(m).__name__
        hs.append(m.register_forward_pre_hook(pre(n)))
        hs.append(m.register_forward_hook(post(n)))
    # inside the two heavy modules: their children
    for m in mods:
        if type(m).__name__ in ('VoxelSetAbstraction', 'PVRCNNHead'):
            for cn, c in m.named_children():
                n = '  %s.%s' % (type(m).__name__[:6], cn)
                hs.append(c.register_forward_pre_hook(pre(n)))
                hs.append(c.register_forward_hook(post(n)))
    # saved tensors of the forward
    saved = {}

    def pack(t):
        if t.is_cuda:
            saved[(t.untyped_storage().data_ptr())] = (t.untyped_storage().nbytes(), tuple(t.shape), str(t.dtype))
        return t
    with torch.autograd.graph.saved_tensors_hooks(pack, lambda t: t):
        step(True)
    for h in hs:
        h.remove()
    tot = sum(v[0] for v in saved.values())
    print('distinct storages saved for backward: %d, %.2f GB; the largest:' % (len(saved), tot / GB))
    for nb, shp, dt in sorted(saved.values(), reverse=True)[:24]:
        print('   %.3f GB  %s %s' % (nb / GB, shp, dt))


if __name__ == '__main__':
    main()
