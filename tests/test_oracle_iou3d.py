"""CPU: pin oracle/iou3d_oracle.c against the reference's own compiled iou3d_cpu.cpp (oracle/_ref, built here from
/root/reference by oracle/build_ref.sh) and against the golden IoU matrix that build produced (travels as .npz)."""
import os

import numpy as np
import pytest

import oracle
from boxes_synth import detection_boxes

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_iou3d.npz')


def test_oracle_iou_matches_golden_from_reference_build():
    g = np.load(G)
    got = oracle.boxes_pairwise(g['a'], g['b'], 1)
    # the reference CPU twin evaluates sin/cos in double and rounds; the .cu (and this oracle) use f32 sin/cos:
    # identical decisions, values within 2e-5
    np.testing.assert_array_equal(got, g['iou'])   # bit-exact with the reference build on this libm
    for th in (0.01, 0.1, 0.7, 0.8):
        safe = np.abs(g['iou'] - th) > 1e-4
        assert ((got > th) == (g['iou'] > th))[safe].all()
    assert (g['iou'] > 0.5).sum() > 30 and (g['iou'] == 0).sum() > 1000


@pytest.mark.skipif(not oracle.have_ref_iou3d(), reason='oracle/_ref not built (needs /root/reference)')
def test_oracle_iou_matches_live_reference_build():
    rng = np.random.default_rng(7)
    a, _ = detection_boxes(rng, 300)
    b, _ = detection_boxes(rng, 200)
    np.testing.assert_allclose(oracle.boxes_pairwise(a, b, 1), oracle.ref_boxes_iou_bev(a, b), rtol=0, atol=2e-5)


def test_iou3d_and_overlap_consistency():
    rng = np.random.default_rng(8)
    a, _ = detection_boxes(rng, 100)
    ov = oracle.boxes_pairwise(a, a, 0)
    iou = oracle.boxes_pairwise(a, a, 1)
    i3 = oracle.boxes_pairwise(a, a, 2)
    area = a[:, 3] * a[:, 4]
    np.testing.assert_allclose(np.diag(ov), area, rtol=2e-4)
    np.testing.assert_allclose(np.diag(iou), 1.0, rtol=2e-4)
    np.testing.assert_allclose(np.diag(i3), 1.0, rtol=2e-4)
    np.testing.assert_allclose(ov, ov.T, rtol=1e-3, atol=1e-3)
    assert (i3 <= iou + 1e-5).all()


def test_nms_greedy_properties():
    rng = np.random.default_rng(9)
    b, s = detection_boxes(rng, 600)
    order = np.argsort(-s, kind='stable')
    bs = b[order]
    keep = oracle.nms(bs, 0.1)
    iou = oracle.boxes_pairwise(bs[keep], bs[keep], 1)
    np.fill_diagonal(iou, 0)
    assert (iou <= 0.1).all()                       # kept boxes do not suppress each other
    full = oracle.boxes_pairwise(bs, bs[keep], 1)
    dropped = np.setdiff1d(np.arange(len(bs)), keep)
    for d in dropped:                                # every dropped box is suppressed by an earlier kept one
        assert (full[d, keep < d] > 0.1).any()
