"""GPU: fused anchor target assignment (csrc/target_assign.hip) — labels bit-exact against the reference golden
(tests/golden/ref_anchor_head.npz) and against the batched torch path on full-size KITTI / Waymo anchor grids."""
import os

import numpy as np
import pytest
import torch

from synth import kitti_batch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_fused_assigner_matches_reference_golden(dev):
    from pcdet.model_cfgs import second_cfg
    from pcdet.models.dense_heads import AnchorHeadSingle
    g = np.load(os.path.join(G, 'ref_anchor_head.npz'))
    head = AnchorHeadSingle(second_cfg().MODEL.DENSE_HEAD, input_channels=24, num_class=3,
                            class_names=['Car', 'Pedestrian', 'Cyclist'], grid_size=np.array([176, 160, 40]),
                            point_cloud_range=np.array([0, -8, -3, 17.6, 8, 1], np.float32)).to(dev)
    out = head.assign_targets(torch.from_numpy(g['head_gt']).to(dev))
    np.testing.assert_array_equal(out['box_cls_labels'].cpu().numpy(), g['head_labels'].astype(np.int32))
    np.testing.assert_allclose(out['box_reg_targets'].cpu().numpy(), g['head_reg_targets'], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(out['reg_weights'].cpu().numpy(), g['head_reg_weights'])


@pytest.mark.parametrize('kind,B', [('kitti', 16), ('waymo', 2)])
def test_fused_assigner_equals_torch_path_full_size(dev, kind, B):
    from pcdet.datasets import SyntheticDataset
    from pcdet.model_cfgs import second_cfg
    from pcdet.models.dense_heads import AnchorHeadSingle
    from pcdet.models.dense_heads.target_assigner import axis_aligned_target_assigner as T
    ds = SyntheticDataset(num_frames=1, kind=kind)
    cfg = second_cfg(kind)
    head = AnchorHeadSingle(cfg.MODEL.DENSE_HEAD, input_channels=8, num_class=3, class_names=cfg.CLASS_NAMES,
                            grid_size=ds.grid_size, point_cloud_range=ds.point_cloud_range).to(dev)
    _, _, gt = kitti_batch(50, B, 2000, waymo=(kind == 'waymo'))
    rng = np.random.default_rng(0)
    gt[..., :2] += rng.normal(0, 0.3, gt[..., :2].shape).astype(np.float32)
    gt[-1, 3:] = 0                                   # ragged: trailing zero rows
    if B > 2:
        gt[1] = 0                                    # a frame without boxes
    gtt = torch.from_numpy(gt).to(dev)
    fused = head.assign_targets(gtt)
    T.FUSED = False
    try:
        ref = head.assign_targets(gtt)
    finally:
        T.FUSED = True
    assert torch.equal(fused['box_cls_labels'], ref['box_cls_labels'])
    assert torch.equal(fused['reg_weights'], ref['reg_weights'])
    torch.testing.assert_close(fused['box_reg_targets'], ref['box_reg_targets'], rtol=1e-6, atol=1e-6)
    assert int((ref['box_cls_labels'] > 0).sum()) > 20 and int((ref['box_cls_labels'] < 0).sum()) > 0
