#!/usr/bin/env python
"""Where one batched CRB stage-2 pass (16 frames, train mode, per-frame BatchNorm statistics) spends its time: wall time per
detector module with a device synchronisation after each, and the RoI-head pieces. Usage: python tools/time_stage2_pass.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

if __name__ == '__main__':
    from pcdet.datasets import SyntheticDataset
    from pcdet.datasets.synthetic import kitti_batch
    from pcdet.model_cfgs import pv_rcnn_cfg
    from pcdet.models import build_network
    from pcdet.utils.frame_bn import per_frame_batchnorm
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = build_network(pv_rcnn_cfg().MODEL, 3, SyntheticDataset(num_frames=2)).to(dev)
    model.train()
    head = model.roi_head
    pts, off, gt = kitti_batch(700, G)
    bidx = np.repeat(np.arange(G, dtype=np.float32), np.diff(off))
    base = {'points': torch.from_numpy(np.concatenate([bidx[:, None], pts], 1)).to(dev),
            'point_frame_offsets': torch.from_numpy(off).to(dev), 'gt_boxes': torch.from_numpy(gt).to(dev), 'batch_size': G,
            'point_frame_counts_host': np.diff(off).tolist()}

    def one(report):
        batch = dict(base)
        t = {}
        with per_frame_batchnorm(model, G), torch.no_grad():
            for mod in model.scheduled_modules():
                if mod is head:
                    break
                torch.cuda.synchronize(); t0 = time.perf_counter()
                batch = mod(batch)
                torch.cuda.synchronize(); t[mod.__class__.__name__] = time.perf_counter() - t0
            batch = dict(batch)
            for name, fn in (('proposal_layer', lambda: head.proposal_layer(batch, nms_config=head.model_cfg.NMS_CONFIG['TRAIN'])),
                             ('assign_targets', lambda: batch.update({'_t': head.assign_targets(batch)}))):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize(); t[name] = time.perf_counter() - t0
            batch['rois'], batch['roi_labels'] = batch['_t']['rois'], batch['_t']['roi_labels']
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pooled = head.roi_grid_pool(batch)
            torch.cuda.synchronize(); t['roi_grid_pool'] = time.perf_counter() - t0
            torch.cuda.synchronize(); t0 = time.perf_counter()
            model.pfe.prefetch_keypoints(dict(base))
            torch.cuda.synchronize(); t['(fps prefetch alone)'] = time.perf_counter() - t0
        with per_frame_batchnorm(model, G):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = pooled.shape[0]
            flat = pooled.permute(0, 2, 1).contiguous().view(n, -1, 1)
            cap = {}
            h = head.shared_fc_layer[4].register_forward_hook(lambda m, i, o: cap.update(a=i[0], z=o))
            shared, rc, rr = head._heads(flat)
            h.remove()
            torch.cuda.synchronize(); t['fc_stack'] = time.perf_counter() - t0
            t0 = time.perf_counter()
            P = n // G
            lab = torch.rand((n, 1), device=dev); tgt = torch.randn((n, 7), device=dev)
            cl, _ = head.get_box_cls_layer_loss({'rcnn_cls': rc, 'rcnn_cls_labels': lab})
            rl = head.get_box_reg_layer_loss({'rcnn_reg': rr, 'reg_sample_targets': tgt})
            tot = float(G) * (cl + rl.mean())
            d, = torch.autograd.grad(tot, cap['z'])
            emb = torch.einsum('gpk,gpj->gkj', d.reshape(G, P, -1), cap['a'].detach().reshape(G, P, -1))
            torch.cuda.synchronize(); t['loss_grad_einsum'] = time.perf_counter() - t0
        if report:
            print('G=%d  ' % G + '  '.join('%s %.1f ms' % (k, 1e3 * v) for k, v in t.items()) + '  | total %.1f ms' % (1e3 * sum(t.values())))
    for k in range(4):
        one(k == 3)
