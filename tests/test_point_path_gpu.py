"""GPU: the PV-RCNN point path of the mirror as a COMPOSITION — VoxelSetAbstraction.forward (keypoint FPS, bilinear BEV
lookup, raw-point SA, four voxel-centre SAs in FEATURES_SOURCE order, concat, fusion layer), StackSAModuleMSG.forward under
each source and PVRCNNHead.roi_grid_pool — against tests/golden/ref_point_path.npz, written by the reference's own classes
(make_goldens.py:gen_point_path: voxel_set_abstraction.py:284-411, pointnet2_modules.py:78-112, pvrcnn_head.py:68-114 with
the PFE / ROI_GRID_POOL sections of the reference's pv_rcnn_active_crb.yaml; pointnet2_stack_cuda answered by the oracle).
The mirror is configured from pcdet.model_cfgs.pv_rcnn_cfg(), so the test also pins that table against the YAML.
Weights are regenerated from the generator's seeded stream by parameter name.

Tolerances (relative to the largest magnitude of the compared array): keypoints exact (FPS picks are indices); eval
features 1e-5 observed -> 1e-4 (eval BatchNorm folded into the 1x1 convs, MFMA f32 accumulation order vs torch's conv2d on
the CPU); train features 3e-5 observed -> 3e-4 (batch statistics summed in another order); gradients 1e-3."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _close(got, ref, rel, what=''):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = float(np.abs(ref).max())
    err = float(np.abs(got - ref).max()) / scale
    assert err <= rel, '%s: max error %.3e of the largest magnitude (tolerance %.0e)' % (what, err, rel)
    return err


class _Level:
    def __init__(self, c, f, dev):
        self.indices, self.features = torch.from_numpy(c.copy()).to(dev), torch.from_numpy(f.copy()).to(dev)


def _build(dev):
    from golden._constants import PP_KEYPOINTS, PP_PCR, PP_VOXEL, point_path_inputs, seeded_state
    from pcdet.model_cfgs import pv_rcnn_cfg
    from pcdet.models.backbones_3d.pfe.voxel_set_abstraction import VoxelSetAbstraction
    from pcdet.models.roi_heads.pvrcnn_head import PVRCNNHead
    g = np.load(os.path.join(GOLD, 'ref_point_path.npz'))
    cfg = pv_rcnn_cfg().MODEL
    cfg.PFE.NUM_KEYPOINTS = PP_KEYPOINTS
    inp = point_path_inputs()
    vsa = VoxelSetAbstraction(cfg.PFE, voxel_size=PP_VOXEL, point_cloud_range=PP_PCR, num_bev_features=inp['bev'].shape[1],
                              num_rawpoint_features=4)
    assert sorted(vsa.state_dict().keys()) == list(g['pp_vsa_keys'])            # the reference's parameter names
    vsa.load_state_dict(seeded_state(vsa, 53))
    head = PVRCNNHead(input_channels=vsa.num_point_features, model_cfg=cfg.ROI_HEAD, num_class=1)
    assert sorted(k for k in head.state_dict() if k.startswith('roi_grid_pool_layer')) == list(g['pp_head_pool_keys'])
    head.load_state_dict(seeded_state(head, 57))
    vsa.to(dev), head.to(dev)

    def batch():
        return {'batch_size': inp['batch_size'], 'points': torch.from_numpy(inp['points'].copy()).to(dev),
                'multi_scale_3d_features': {k: _Level(c, f, dev) for k, (c, f) in inp['levels'].items()},
                'spatial_features': torch.from_numpy(inp['bev'].copy()).to(dev), 'spatial_features_stride': 8,
                'rois': torch.from_numpy(inp['rois'].copy()).to(dev)}
    return vsa, head, batch, inp, g


def _run(vsa, head, bd, inp, dev):
    bd = vsa(bd)
    bd['point_cls_scores'] = torch.from_numpy(inp['scores'].copy()).to(dev)
    return bd, head.roi_grid_pool(bd)


@pytest.mark.parametrize('channels_last', [False, True])
def test_point_path_eval_matches_reference_classes(dev, channels_last):
    """inference path of the mirror: fused group + MLP + max kernel, folded BatchNorm, two-radius ball query, grouped RoI
    queries; the BEV map in both storage orders the mirror produces"""
    vsa, head, batch, inp, g = _build(dev)
    vsa.eval(), head.eval()
    bd = batch()
    if channels_last:
        bd['spatial_features'] = bd['spatial_features'].contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        bd, pooled = _run(vsa, head, bd, inp, dev)
    np.testing.assert_array_equal(bd['point_coords'].cpu().numpy(), g['pp_point_coords'])
    e1 = _close(bd['point_features_before_fusion'], g['pp_eval_before_fusion'], 1e-4, 'before_fusion')
    e2 = _close(bd['point_features'], g['pp_eval_point_features'], 1e-4, 'point_features')
    e3 = _close(pooled, g['pp_eval_pooled'], 1e-4, 'roi_grid_pool')
    # per feature source (concat order of voxel_set_abstraction.py:349-404: bev | raw | x_conv1..4), so a swapped pair of
    # equally wide sources cannot hide behind the global maximum
    got, ref = bd['point_features_before_fusion'].cpu().numpy(), g['pp_eval_before_fusion']
    col = 0
    for name, w in (('bev', inp['bev'].shape[1]), ('raw', 32), ('x_conv1', 32), ('x_conv2', 64), ('x_conv3', 128), ('x_conv4', 128)):
        _close(got[:, col:col + w], ref[:, col:col + w], 1e-4, name)
        col += w
    assert col == ref.shape[1]
    print('eval errors: before_fusion %.2e point_features %.2e pooled %.2e' % (e1, e2, e3))


def test_point_path_eval_module_route_matches_reference_classes(dev, monkeypatch):
    """the unfused route (QueryAndGroup + the nn.Sequential MLPs, what a config outside the fused kernel's shapes takes)"""
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_modules as pm
    monkeypatch.setattr(pm, 'FUSED_SA_EVAL', False)
    monkeypatch.setattr(pm, 'ROWS_TRAIN', False)
    vsa, head, batch, inp, g = _build(dev)
    vsa.eval(), head.eval()
    with torch.no_grad():
        bd, pooled = _run(vsa, head, batch(), inp, dev)
    _close(bd['point_features_before_fusion'], g['pp_eval_before_fusion'], 1e-4, 'before_fusion')
    _close(pooled, g['pp_eval_pooled'], 1e-4, 'roi_grid_pool')


def test_point_path_train_matches_reference_classes_and_gradients(dev):
    """training path of the mirror (split first layer, row BatchNorm kernels, fused BN+ReLU+max+concat) with batch
    statistics; backward through the grouped-first-layer gradient and the arg-max scatter"""
    vsa, head, batch, inp, g = _build(dev)
    vsa.train(), head.train()
    bd, pooled = _run(vsa, head, batch(), inp, dev)
    e1 = _close(bd['point_features_before_fusion'], g['pp_train_before_fusion'], 3e-4, 'before_fusion')
    e2 = _close(bd['point_features'], g['pp_train_point_features'], 3e-4, 'point_features')
    e3 = _close(pooled, g['pp_train_pooled'], 3e-4, 'roi_grid_pool')
    (pooled.square().sum() + bd['point_features'].square().sum()).backward()
    params = dict(vsa.named_parameters())
    errs = []
    for k in g.files:
        if not k.startswith('pp_grad/'):
            continue
        n = k[len('pp_grad/'):]
        p = head.roi_grid_pool_layer.mlps[0][0].weight if n.startswith('roi_grid_pool_layer') else params[n]
        errs.append(_close(p.grad, g[k], 1e-3, n))
    assert len(errs) == 5
    _close(vsa.vsa_point_feature_fusion[1].running_mean, g['pp_running_mean_after'], 3e-4, 'running_mean')
    print('train errors: before_fusion %.2e point_features %.2e pooled %.2e grads %s' % (e1, e2, e3, ['%.1e' % e for e in errs]))


def test_stack_sa_module_alone_with_ragged_counts_matches_reference(dev):
    """StackSAModuleMSG.forward (pointnet2_modules.py:78-112) on the x_conv2 source: 40 queries in frame 0, 7 in frame 1"""
    from golden._constants import PP_PCR, PP_VOXEL
    from pcdet.utils import common_utils
    vsa, head, batch, inp, g = _build(dev)
    sa = vsa.SA_layers[1].eval()
    c, f = inp['levels']['x_conv2']
    xyz = common_utils.get_voxel_centers(torch.from_numpy(c[:, 1:4].copy()).to(dev), 2, PP_VOXEL, PP_PCR)
    cnt = torch.from_numpy(np.bincount(c[:, 0], minlength=2).astype(np.int32)).to(dev)
    with torch.no_grad():
        _, y = sa(xyz=xyz.contiguous(), xyz_batch_cnt=cnt, new_xyz=torch.from_numpy(g['pp_sa_queries']).to(dev),
                  new_xyz_batch_cnt=torch.tensor([40, 7], dtype=torch.int32, device=dev),
                  features=torch.from_numpy(f.copy()).to(dev))
    _close(y, g['pp_sa_out'], 1e-4, 'sa_out')
