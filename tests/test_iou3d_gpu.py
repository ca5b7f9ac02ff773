"""GPU parity: rotated BEV overlap / IoU / 3-D IoU and NMS (csrc/iou3d_nms.hip through the C-ABI and the
pcdet.ops.iou3d_nms.iou3d_nms_utils mirror) vs the oracle (pinned bit-exact to the reference's compiled iou3d_cpu.cpp).
IoU values: |hip - oracle| <= 2e-5 (f32 sin/cos/atan2 of a different libm). NMS picks: identical index lists on
margin-aware inputs (no pair within 1e-4 of the threshold)."""
import numpy as np
import pytest
import torch

import oracle
from boxes_synth import detection_boxes

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_pairwise_modes(dev):
    from pcdet.ops.iou3d_nms import iou3d_nms_utils as U
    rng = np.random.default_rng(1)
    a, _ = detection_boxes(rng, 700)
    b, _ = detection_boxes(rng, 333)
    b[:100] = a[:100] + rng.normal(0, 0.1, (100, 7)).astype(np.float32)
    for mode, fn in ((0, U.boxes_overlap_bev), (1, U.boxes_iou_bev), (2, U.boxes_iou3d_gpu)):
        got = fn(_t(a, dev), _t(b, dev)).cpu().numpy()
        ref = oracle.boxes_pairwise(a, b, mode)
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5 * max(1.0, ref.max()))
    g = np.load(__file__.replace('test_iou3d_gpu.py', 'golden/ref_iou3d.npz'))
    got = U.boxes_iou_bev(_t(g['a'], dev), _t(g['b'], dev)).cpu().numpy()
    np.testing.assert_allclose(got, g['iou'], rtol=0, atol=2e-5)      # vs the reference's own compiled CPU code
    np.testing.assert_allclose(U.boxes_bev_iou_cpu(g['a'], g['b']), g['iou'], rtol=0, atol=2e-5)


def _margin_safe(boxes_sorted, thresh):
    iou = oracle.boxes_pairwise(boxes_sorted, boxes_sorted, 1)
    bad = np.abs(iou - thresh) < 1e-4
    np.fill_diagonal(bad, False)
    drop = np.unique(np.nonzero(bad)[0])
    return np.delete(np.arange(len(boxes_sorted)), drop)


@pytest.mark.parametrize('n,thresh', [(1024, 0.7), (4096, 0.1), (9000, 0.8), (100, 0.01), (65, 0.5), (1, 0.5)])
def test_nms_gpu_picks_identical(dev, n, thresh):
    from pcdet.ops.iou3d_nms import iou3d_nms_utils as U
    rng = np.random.default_rng(n)
    b, s = detection_boxes(rng, n, n_obj=max(2, n // 12))
    s = ((rng.permutation(n) + 1) / np.float32(n + 1)).astype(np.float32)   # distinct scores -> unique sort order
    assert len(np.unique(s)) == n
    order = np.argsort(-s, kind='stable')
    ok = _margin_safe(b[order], thresh) if n <= 4096 else np.arange(n)
    b, s = b[order][ok], s[order][ok]
    keep, _ = U.nms_gpu(_t(b, dev), _t(s, dev), thresh)
    ref = oracle.nms(b, thresh)                                       # b is already score sorted
    if n > 4096:          # too many pairs to pre-filter: allow only margin-explained differences
        assert len(keep) > 0
        k = keep.cpu().numpy()
        if not np.array_equal(k, ref):
            iou = oracle.boxes_pairwise(b[np.union1d(k, ref)], b[np.union1d(k, ref)], 1)
            assert (np.abs(iou - thresh) < 1e-4).any()
    else:
        np.testing.assert_array_equal(keep.cpu().numpy(), ref)
    # pre_maxsize + scores unsorted input order: indices refer to the INPUT
    perm = rng.permutation(len(b))
    keep2, _ = U.nms_gpu(_t(b[perm], dev), _t(s[perm], dev), thresh, pre_maxsize=min(512, len(b)))
    ref2 = oracle.nms(b[:min(512, len(b))], thresh)
    inv = np.argsort(perm)
    if n <= 4096:
        np.testing.assert_array_equal(keep2.cpu().numpy(), inv[ref2])


def test_nms_normal_and_batched_padded(dev):
    from pcdet.ops.iou3d_nms import iou3d_nms_utils as U
    rng = np.random.default_rng(3)
    B, N = 5, 700
    boxes = np.zeros((B, N, 7), np.float32)
    counts = np.array([700, 1, 0, 333, 64], np.int32)
    refs = []
    for f in range(B):
        b, s = detection_boxes(rng, N)
        boxes[f] = b[np.argsort(-s, kind='stable')]
        refs.append(oracle.nms(boxes[f, :counts[f]], 0.3, rotated=False))
    keep, num = U.nms_batched(_t(boxes, dev), _t(counts, dev), 0.3, 128, rotated=False)
    keep, num = keep.cpu().numpy(), num.cpu().numpy()
    for f in range(B):
        r = refs[f][:128]
        assert num[f] == len(r)
        np.testing.assert_array_equal(keep[f, :len(r)], r)
        assert (keep[f, len(r):] == -1).all()
    k1, _ = U.nms_normal_gpu(_t(boxes[0], dev), torch.arange(N, 0, -1, device=dev).float(), 0.3)
    np.testing.assert_array_equal(k1.cpu().numpy(), refs[0])
    idx, valid = U.nms_gpu_padded(_t(boxes[3, :333], dev), torch.arange(333, 0, -1, device=dev).float(), 0.3, None, 50,
                                  rotated=True)
    r = oracle.nms(boxes[3, :333], 0.3)[:50]
    np.testing.assert_array_equal(idx.cpu().numpy()[valid.cpu().numpy()], r)


def test_batched_nms_prefix_stage_gives_the_full_scan_picks(dev):
    """max_keep << nmax (training proposals: 512 of 9,000): crb_nms_batched first scans the leading 1,024 boxes and computes the
    full suppression matrix only for frames that have not reached max_keep there. Three frames in one call: (0) spread boxes —
    final after the prefix stage; (1) ~25 objects with 160 jittered copies each — far fewer than max_keep survive the prefix,
    the full stage decides; (2) fewer boxes than the prefix. Picks identical to the oracle's sequential NMS."""
    from pcdet.ops.iou3d_nms import iou3d_nms_utils as U
    rng = np.random.default_rng(11)
    N, keep_n, thresh = 4096, 128, 0.3
    frames = []
    b0, s0 = detection_boxes(rng, N, n_obj=1500)
    frames.append(b0[np.argsort(-s0, kind='stable')])
    centres, _ = detection_boxes(rng, 25, n_obj=25)
    b1 = np.repeat(centres, 164, axis=0)[:N].copy()
    b1[:, :2] += rng.normal(0, 0.15, (N, 2)).astype(np.float32)
    b1[:, 6] += rng.normal(0, 0.05, N).astype(np.float32)
    frames.append(b1[rng.permutation(N)])
    b2, s2 = detection_boxes(rng, N, n_obj=300)
    frames.append(b2[np.argsort(-s2, kind='stable')])
    counts = np.array([N, N, 900], np.int32)
    boxes = np.zeros((3, N, 7), np.float32)
    refs = []
    for f, b in enumerate(frames):
        ok = _margin_safe(b[:counts[f]], thresh)
        bb = b[:counts[f]][ok]
        counts[f] = len(bb)
        boxes[f, :len(bb)] = bb
        refs.append(oracle.nms(bb, thresh)[:keep_n])
    assert len(refs[0]) == keep_n and refs[0][-1] < 1024            # frame 0 is decided inside the prefix
    assert len(refs[1]) < keep_n or refs[1][-1] >= 1024            # frame 1 needs rows beyond it
    keep, num = U.nms_batched(_t(boxes, dev), _t(counts, dev), thresh, keep_n, rotated=True)
    keep, num = keep.cpu().numpy(), num.cpu().numpy()
    for f in range(3):
        assert num[f] == len(refs[f]), (f, num[f], len(refs[f]))
        np.testing.assert_array_equal(keep[f, :num[f]], refs[f])
        assert (keep[f, num[f]:] == -1).all()


def test_batched_nms_at_the_training_proposal_size_prefix_equals_single_stage(dev):
    """9,000 score-sorted boxes per frame, 512 kept, threshold 0.8 (the training proposal layer of PV-RCNN): the prefix stage must
    return the picks of the single-stage scan — obtained from the same entry point with max_keep = N, which switches the prefix
    stage off — for frames with few and with many overlaps"""
    from pcdet.ops.iou3d_nms import iou3d_nms_utils as U
    rng = np.random.default_rng(21)
    N = 9000
    boxes = np.zeros((3, N, 7), np.float32)
    for f, n_obj in enumerate((3000, 150, 30)):
        b, s = detection_boxes(rng, N, n_obj=n_obj)
        boxes[f] = b[np.argsort(-s, kind='stable')]
    counts = np.array([N, N, 7000], np.int32)
    kp, nump = U.nms_batched(_t(boxes, dev), _t(counts, dev), 0.8, 512, rotated=True)
    kf, numf = U.nms_batched(_t(boxes, dev), _t(counts, dev), 0.8, N, rotated=True)
    kp, nump, kf, numf = kp.cpu().numpy(), nump.cpu().numpy(), kf.cpu().numpy(), numf.cpu().numpy()
    for f in range(3):
        k = min(512, numf[f])
        assert nump[f] == k
        np.testing.assert_array_equal(kp[f, :k], kf[f, :k])


def test_model_nms_utils_contract(dev):
    """class_agnostic_nms / multi_classes_nms (model_nms_utils.py:6-66): selected indices refer to the INPUT arrays, scores come
    back with them, per-class results are concatenated in class order — checked against the oracle NMS applied to the
    thresholded, score-sorted, NMS_PRE_MAXSIZE-capped boxes; plus the empty cases"""
    from pcdet.config import EasyDict
    from pcdet.models.model_utils import model_nms_utils as M
    rng = np.random.default_rng(5)
    n = 600
    b, _ = detection_boxes(rng, n, n_obj=40)
    s = ((rng.permutation(n) + 1) / np.float32(n + 1)).astype(np.float32)
    cfg = EasyDict({'NMS_TYPE': 'nms_gpu', 'NMS_THRESH': 0.3, 'NMS_PRE_MAXSIZE': 256, 'NMS_POST_MAXSIZE': 50})

    def expect(scores, thresh):
        cand = np.nonzero(scores >= thresh)[0] if thresh is not None else np.arange(len(scores))
        cand = cand[np.argsort(-scores[cand], kind='stable')][:cfg.NMS_PRE_MAXSIZE]
        return cand[oracle.nms(b[cand], cfg.NMS_THRESH)][:cfg.NMS_POST_MAXSIZE]

    for thresh in (None, 0.35):
        sel, sc = M.class_agnostic_nms(_t(s, dev), _t(b, dev), cfg, score_thresh=thresh)
        want = expect(s, thresh)
        np.testing.assert_array_equal(sel.cpu().numpy(), want)
        np.testing.assert_array_equal(sc.cpu().numpy(), s[want])
    sel, sc = M.class_agnostic_nms(_t(s, dev), _t(b, dev), cfg, score_thresh=2.0)         # nothing passes
    assert len(sel) == 0 and len(sc) == 0
    cls = np.stack([s, s[::-1].copy(), np.zeros_like(s)], 1)
    ps, pl, pb = M.multi_classes_nms(_t(cls, dev), _t(b, dev), cfg, score_thresh=0.5)
    w0, w1 = expect(cls[:, 0], 0.5), expect(cls[:, 1], 0.5)
    np.testing.assert_array_equal(pl.cpu().numpy(), np.r_[np.zeros(len(w0), np.int64), np.ones(len(w1), np.int64)])
    np.testing.assert_array_equal(ps.cpu().numpy(), np.r_[cls[w0, 0], cls[w1, 1]])
    np.testing.assert_array_equal(pb.cpu().numpy(), np.r_[b[w0], b[w1]])


def _oracle_nms_first(b, thresh, keep_n):
    """the oracle's greedy NMS stopped after keep_n picks (pairwise IoU from the pinned oracle, one candidate against the kept set)"""
    kept = []
    for i in range(len(b)):
        if kept and (oracle.boxes_pairwise(b[i:i + 1], b[kept], 1)[0] > thresh).any():
            continue
        kept.append(i)
        if len(kept) == keep_n:
            break
    return np.asarray(kept, np.int64)


def test_nms_picks_on_unfiltered_training_size_proposal_sets_counted(dev):
    """how often do the picks differ from the oracle's when NOTHING is filtered? 100 seeded sets of 9,000
    score-sorted proposals (the training NMS: threshold 0.8, 512 kept) straight from the box generator - 80 sets of loosely
    scattered boxes (most survive) and 20 of tight clusters around 60 objects (half of the candidates are suppressed, IoUs inside
    a cluster lie around the threshold). The first 512 picks are compared as they are; every difference has to be explained by a pair of
    boxes whose IoU lies within 1e-4 of the threshold (f32 sin / cos of a different libm on the two sides). The test prints how
    many sets and boxes differ - the number README / DESIGN section 4 quote next to "bit-exact on margin-safe inputs"."""
    from pcdet.ops.iou3d_nms import iou3d_nms_utils as U
    thresh, keep_n, n = 0.8, 512, 9000
    differing, boxes_diff = 0, 0
    for seed in range(100):
        rng = np.random.default_rng(70000 + seed)
        if seed < 80:
            b, s = detection_boxes(rng, n, n_obj=n // 12)
        else:             # proposal-like: 150 near-copies of each of 60 boxes (IoU around the threshold inside a cluster)
            c, _ = detection_boxes(rng, 60, n_obj=60)
            b = np.repeat(c, n // 60, axis=0).copy()
            b[:, :2] += rng.normal(0, 0.12, (n, 2)).astype(np.float32)
            b[:, 3:6] *= rng.uniform(0.95, 1.05, (n, 3)).astype(np.float32)
            b[:, 6] += rng.normal(0, 0.03, n).astype(np.float32)
            s = rng.uniform(0, 1, n).astype(np.float32)
        b = b[np.argsort(-s, kind='stable')]
        keep, _ = U.nms_gpu(_t(b, dev), torch.arange(n, 0, -1, device=dev).float(), thresh)
        k = keep.cpu().numpy()[:keep_n]
        ref = _oracle_nms_first(b, thresh, keep_n)
        if not np.array_equal(k, ref):
            differing += 1
            sym = np.setxor1d(k, ref)
            boxes_diff += len(sym)
            first = int(min(sym))                        # the first box the two sides disagree on: one of its pairs is in the margin
            iou = oracle.boxes_pairwise(b[first:first + 1], b[:first], 1)[0]
            assert (np.abs(iou - thresh) < 1e-4).any(), (seed, first)
    print('NMS picks, 100 unfiltered sets of %d proposals (threshold %.1f, first %d picks): %d sets differ from the oracle, %d '
          'boxes in the symmetric difference in total' % (n, thresh, keep_n, differing, boxes_diff))
    assert differing <= 10
