"""CPU: oracle/anchor_head_oracle.py (the independent dense half used by smoke() / second_step_cpu) pinned by the goldens
the reference's own AnchorHeadSingle / BaseBEVBackbone produced (tests/golden/ref_anchor_head.npz, ref_bev_vfe.npz)."""
import os
import types

import numpy as np
import torch

from oracle import anchor_head_oracle as aho
from golden._constants import small_head_cfg

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_oracle_anchors_targets_and_losses_match_reference_golden():
    g = np.load(os.path.join(GOLD, 'ref_anchor_head.npz'))
    cfgs = small_head_cfg()['ANCHOR_GENERATOR_CONFIG']
    anchors = aho.generate_anchors([0, -8, -3, 17.6, 8, 1], cfgs, [np.array([22, 20])] * 3)
    np.testing.assert_array_equal(np.stack([a.numpy() for a in anchors]), g['head_anchors'])
    gt = torch.from_numpy(g['head_gt'])
    labels, targets, weights = aho.assign_targets(anchors, gt, ['Car', 'Pedestrian', 'Cyclist'], cfgs)
    np.testing.assert_array_equal(labels.numpy(), g['head_labels'].astype(np.int32))
    np.testing.assert_allclose(targets.numpy(), g['head_reg_targets'], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(weights.numpy(), g['head_reg_weights'])
    # head + losses + gradient with the reference's weights
    st = {k[len('head_state/'):]: torch.from_numpy(g[k]) for k in g.files if k.startswith('head_state/')}
    mk = lambda n: types.SimpleNamespace(weight=st['conv_%s.weight' % n], bias=st['conv_%s.bias' % n])
    head = types.SimpleNamespace(conv_cls=mk('cls'), conv_box=mk('box'), conv_dir_cls=mk('dir_cls'))
    feats = torch.from_numpy(g['head_feats']).requires_grad_(True)
    cls, box, dr = aho.head_preds(head, feats)
    total, lc, ll, ld = aho.rpn_loss(cls, box, dr, labels, targets, anchors)
    np.testing.assert_allclose([float(total), float(lc), float(ll), float(ld)], g['head_loss'], rtol=2e-6)
    total.backward()
    np.testing.assert_allclose(feats.grad.numpy(), g['head_feats_grad'], rtol=1e-4, atol=1e-8)


def test_oracle_bev_backbone_matches_reference_golden():
    from pcdet.config import EasyDict
    from pcdet.models.backbones_2d.base_bev_backbone import BaseBEVBackbone
    g = np.load(os.path.join(GOLD, 'ref_bev_vfe.npz'))
    cfg = EasyDict({'LAYER_NUMS': [2, 1], 'LAYER_STRIDES': [1, 2], 'NUM_FILTERS': [8, 16], 'UPSAMPLE_STRIDES': [1, 2],
                    'NUM_UPSAMPLE_FILTERS': [8, 8]})
    m = BaseBEVBackbone(cfg, input_channels=6)                 # only a parameter container here: the oracle walks the layer list
    m.load_state_dict({k[len('bev_state/'):]: torch.from_numpy(g[k]) for k in g.files if k.startswith('bev_state/')})
    y = aho.bev_backbone(m, torch.from_numpy(g['bev_x']))
    np.testing.assert_allclose(y.detach().numpy(), g['bev_y'], rtol=1e-4, atol=1e-5)
