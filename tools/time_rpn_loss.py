#!/usr/bin/env python
"""device time of the RPN loss (AnchorHeadTemplate.get_loss: focal classification + smooth-L1 box + direction loss over
16 x 211,200 anchors) forward + backward with respect to the head outputs, and of target assignment, at the bench shapes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np, torch

if __name__ == '__main__':
    from pcdet.datasets import SyntheticDataset
    from pcdet.datasets.synthetic import kitti_batch
    from pcdet.model_cfgs import second_cfg
    from pcdet.models import build_network
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    B = 16
    ds = SyntheticDataset(num_frames=B)
    model = build_network(second_cfg().MODEL, 3, ds).to(dev).train()
    head = model.dense_head
    pts, off, gt = kitti_batch(0, B)
    gtb = torch.from_numpy(gt).to(dev)
    cls = torch.randn(B, 200, 176, 18, device=dev, requires_grad=True)
    box = (torch.randn(B, 200, 176, 42, device=dev) * 0.3).requires_grad_(True)
    dr = torch.randn(B, 200, 176, 12, device=dev, requires_grad=True)

    def once(assign=True):
        if assign:
            t = head.assign_targets(gt_boxes=gtb)
            head.forward_ret_dict.update(t)
        head.forward_ret_dict.update({'cls_preds': cls, 'box_preds': box, 'dir_cls_preds': dr})
        loss, tb = head.get_loss()
        loss.backward()
        cls.grad = box.grad = dr.grad = None
        return loss
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    for assign in (True, False):
        ts = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); once(assign); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print('RPN loss fwd+bwd %s target assignment: %.3f ms (median of 10)' % ('with' if assign else 'without', float(np.median(ts))))
