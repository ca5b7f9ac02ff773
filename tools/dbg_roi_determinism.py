#!/usr/bin/env python
"""which op of the PV-RCNN RoI head (eval) is not bit-reproducible? tests/test_pvrcnn_gpu.py saw rcnn_cls /
rcnn_reg 1.6e-8 apart ONCE in a full-suite run with rois / point features bit-equal. Runs the chain up to the RoI head once, then
repeats each stage of the head on the SAME inputs and counts the distinct bit patterns per stage:
  pool      roi_grid_pool (grouped ball query + sa_mlp2_max_kernel: ours)
  fc0       the 27,648 -> 256 layer as torch.addmm (vendor GEMM, K = 27,648: split-K candidates)
  tail      the remaining folded conv1d layers + cls / reg branches (vendor GEMMs, small)
usage: python tools/dbg_roi_determinism.py [repeats]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import hashlib  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402


def digest(t):
    return hashlib.sha1(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:12]


if __name__ == '__main__':
    from pcdet.datasets import SyntheticDataset
    from pcdet.datasets.synthetic import kitti_batch
    from pcdet.model_cfgs import pv_rcnn_cfg
    from pcdet.models import build_network
    from pcdet.utils.fold_utils import fold_conv_bn
    from pcdet.models.roi_heads.pvrcnn_head import _gc_order
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = build_network(pv_rcnn_cfg().MODEL, 3, SyntheticDataset(num_frames=2)).to(dev).eval()
    B = 2
    pts, off, gt = kitti_batch(7, B, 20000)
    bidx = np.repeat(np.arange(B, dtype=np.float32), np.diff(off))
    b = {'points': torch.from_numpy(np.concatenate([bidx[:, None], pts], 1)).to(dev), 'point_frame_offsets': torch.from_numpy(off).to(dev),
         'gt_boxes': torch.from_numpy(gt).to(dev), 'batch_size': B, 'point_frame_counts_host': np.diff(off).tolist()}
    head = model.roi_head
    with torch.no_grad():
        model.pfe.prefetch_keypoints(b)
        for mod in model.scheduled_modules():
            if mod is head:
                break
            b = mod(b)
        head.proposal_layer(b, nms_config=head.model_cfg.NMS_CONFIG['TEST'])
        pools = [head.roi_grid_pool(b) for _ in range(reps)]
        print('pool   : %d distinct of %d' % (len({digest(p) for p in pools}), reps), flush=True)
        pooled = pools[0]
        n, g3, c = pooled.shape
        mods = list(head.shared_fc_layer)
        w0, b0 = fold_conv_bn(mods[0], mods[1], _gc_order(c, g3))
        flat = pooled.reshape(n, g3 * c)
        outs = [torch.addmm(b0, flat, w0.t()) for _ in range(reps * 5)]
        ds = [digest(o) for o in outs]
        print('fc0    : %d distinct of %d (M=%d K=%d N=%d)' % (len(set(ds)), len(ds), n, g3 * c, w0.shape[0]), flush=True)
        if len(set(ds)) > 1:
            ref = outs[0]
            print('         largest difference between two runs: %.3e (scale %.3e)' %
                  (max(float((o - ref).abs().max()) for o in outs), float(ref.abs().max())))
        # the same GEMM at the scoring batch (16 frames x 128 RoIs)
        big = flat.repeat(8, 1)
        outs = [torch.addmm(b0, big, w0.t()) for _ in range(reps)]
        print('fc0@16 : %d distinct of %d (M=%d)' % (len({digest(o) for o in outs}), reps, big.shape[0]), flush=True)
        tails = [head._heads_eval(pooled, 1)[0] for _ in range(reps)]
        for k, name in enumerate(('shared', 'rcnn_cls', 'rcnn_reg')):
            print('%-7s: %d distinct of %d' % (name, len({digest(t[k]) for t in tails}), reps), flush=True)
        import torch.cuda.tunable as tunable
        print('TunableOp enabled: %s, tuning: %s' % (tunable.is_enabled(), tunable.tuning_is_enabled()))
