"""Device-side CRB scoring primitives shared by the strategies (MI355X redesign of the hot loops of
pcdet/query_strategies/crb_sampling.py): fixed-stride per-frame records, frame sharding + RCCL all-gather, the stage-3
density prior and the HIP greedy selection."""
import numpy as np
import torch
import torch.distributed as dist

from crbhip import lib, check, ptr, cur_stream, require_cuda

GT_STAT_FIELDS = 5                                  # num_bbox, n_counted, mean, median, variance per class


class RecordLayout:
    """fixed-stride f32 row per pool frame — what every rank all-gathers:
         entropy | n_box | labels[MB] | density[MB] | rcnn_cls[MB] | rcnn_reg[7 MB] | gt_stats[5 C]
    MB = the largest number of boxes a frame can carry: the final NMS keeps at most
    min(POST_PROCESSING NMS_POST_MAXSIZE, NMS_PRE_MAXSIZE, #candidates) boxes, the MC-dropout means have one entry per RoI
    (ROI_HEAD TEST NMS_POST_MAXSIZE). gt_stats are the per-class GT point statistics the caller pickles after every
    query (Strategy.save_points / save_active_labels), so they travel with the scores."""

    def __init__(self, max_box=128, num_class=3):
        self.max_box, self.num_class = int(max_box), int(num_class)
        mb = self.max_box
        self.o_labels = 2
        self.o_density = 2 + mb
        self.o_cls = 2 + 2 * mb
        self.o_reg = 2 + 3 * mb
        self.o_gt = 2 + 10 * mb
        self.stride = self.o_gt + GT_STAT_FIELDS * self.num_class

    @classmethod
    def for_model(cls, model):
        cfg = model.model_cfg
        nc = len(cfg.DENSE_HEAD.ANCHOR_GENERATOR_CONFIG)
        nms = cfg.POST_PROCESSING.NMS_CONFIG
        post = min(int(nms.NMS_POST_MAXSIZE), int(nms.NMS_PRE_MAXSIZE))
        roi = cfg.get('ROI_HEAD', None)
        if roi is not None:
            r = int(roi.NMS_CONFIG.TEST.NMS_POST_MAXSIZE)
            post = min(post, r)                  # the RoI head hands exactly r boxes per frame to post-processing
            return cls(max(post, r), nc)
        return cls(post, nc)

    def __eq__(self, other):
        return isinstance(other, RecordLayout) and (self.max_box, self.num_class) == (other.max_box, other.num_class)


DEFAULT_LAYOUT = RecordLayout(128, 3)               # pv_rcnn_active_crb.yaml: 128 RoIs, 3 classes
MAX_BOX = DEFAULT_LAYOUT.max_box
REC_STRIDE = DEFAULT_LAYOUT.stride


def pack_records(rec, layout=None):
    """crb_frame_records(...) dict -> (B, layout.stride) f32"""
    L = layout or DEFAULT_LAYOUT
    B = rec['entropy'].shape[0]
    P = rec['pred_labels'].shape[1]
    if P > L.max_box:
        raise ValueError('record layout holds %d boxes per frame but post-processing produced %d: build the layout with '
                         'RecordLayout.for_model(model)' % (L.max_box, P))
    out = torch.zeros((B, L.stride), dtype=torch.float32, device=rec['entropy'].device)
    out[:, 0] = rec['entropy']
    out[:, 1] = rec['num'].float()
    out[:, L.o_labels:L.o_labels + P] = rec['pred_labels'].float()
    out[:, L.o_density:L.o_density + P] = rec['density']
    if rec['batch_rcnn_cls'] is not None:
        R = rec['batch_rcnn_cls'].shape[1]
        if R > L.max_box:
            raise ValueError('record layout holds %d RoIs per frame, the head produced %d' % (L.max_box, R))
        out[:, L.o_cls:L.o_cls + R] = rec['batch_rcnn_cls'].reshape(B, R)
        out[:, L.o_reg:L.o_reg + 7 * R] = rec['batch_rcnn_reg'].reshape(B, 7 * R)
    if rec.get('gt_stats', None) is not None:
        out[:, L.o_gt:] = rec['gt_stats'].reshape(B, -1)
    return out


def unpack_records(records, layout=None):
    """(F, layout.stride) -> dict of views"""
    L = layout or DEFAULT_LAYOUT
    mb = L.max_box
    return {'entropy': records[:, 0], 'num': records[:, 1].long(),
            'labels': records[:, L.o_labels:L.o_labels + mb].long(),
            'density': records[:, L.o_density:L.o_density + mb],
            'rcnn_cls': records[:, L.o_cls:L.o_cls + mb].unsqueeze(-1),
            'rcnn_reg': records[:, L.o_reg:L.o_reg + 7 * mb].reshape(-1, mb, 7),
            'gt_stats': records[:, L.o_gt:].reshape(-1, L.num_class, GT_STAT_FIELDS)}


def shard_indices(n, rank, world):
    """rank-strided shard with wrap-around padding — the reference's eval DistributedSampler
    (pcdet/datasets/__init__.py:37-46) — -> (indices of this rank, per-rank count)"""
    per = (n + world - 1) // world
    total = per * world
    idx = list(range(n)) + list(range(total - n))
    return idx[rank:total:world], per


# diagnostics of the collectives of one process: every all_gather_rows call appends (rows per rank, bytes per rank, seconds
# from "this rank's rows are ready" to "gathered rows are here" — i.e. wait for the slowest rank + the transfer). The timing
# synchronises the device; it is switched on by bench.py / tests only (COLLECTIVE_LOG = []), never in the product path.
COLLECTIVE_LOG = None


# CRB_FORCE_DIST=1: the collectives run at world size 1 too (a world-size-1 `nccl` group on one GPU loads RCCL and executes them:
# tests/test_multirank_gpu.py); results are the same rows
FORCE_COLLECTIVE = __import__('os').environ.get('CRB_FORCE_DIST', '0') == '1'


def all_gather_rows(local, n_total, world, backend_device=None):
    """local (per, S) rows of this rank (rank-strided shard) -> (n_total, S) rows in dataset order, on every rank.
    One all_gather_into_tensor (RCCL over xGMI on the GPU node; gloo in the CPU tests)."""
    if world == 1 and not FORCE_COLLECTIVE:
        return local[:n_total]
    per = local.shape[0]
    log = COLLECTIVE_LOG
    if log is not None:
        import time
        if local.is_cuda:
            torch.cuda.synchronize(local.device)
        t0 = time.perf_counter()
    gathered = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if local.is_cuda and dist.get_backend() == 'gloo':       # gloo dry runs on GPU tensors: gather through the host
        g = torch.empty(gathered.shape, dtype=local.dtype)
        dist.all_gather_into_tensor(g, local.contiguous().cpu())
        gathered.copy_(g)
    else:
        dist.all_gather_into_tensor(gathered, local.contiguous())
    if log is not None:
        if local.is_cuda:
            torch.cuda.synchronize(local.device)
        log.append({'rows_per_rank': per, 'bytes_per_rank': local.numel() * local.element_size(),
                    'seconds': time.perf_counter() - t0, 'backend': dist.get_backend()})
    # gathered[r*per + k] = global index k*world + r  -> interleave back and drop the wrap-around padding
    out = gathered.view(world, per, *local.shape[1:]).transpose(0, 1).reshape(world * per, *local.shape[1:])
    return out[:n_total]


def density_prior(density_all, label_all, num_class, alpha=0.95):
    """per-class evaluation axis and uniform prior over the central `alpha` density interval
    (crb_sampling.py:250-260). -> xaxis (C,400) f64, prior (C,400) f64 numpy"""
    xaxis = np.zeros((num_class, 400), np.float64)
    prior = np.zeros((num_class, 400), np.float64)
    for c in range(num_class):
        d = torch.sort(density_all[label_all == (c + 1)])[0]
        n = d.numel()
        if n == 0:
            # no predicted box of this class anywhere in the pool: the reference indexes an empty tensor and crashes
            # (crb_sampling.py:255-258); here the class simply never contributes (every candidate has 0 boxes of it)
            xaxis[c] = np.linspace(-50, 50, 400)
            continue
        gmax = int(d[-1])
        ghigh = int(d[int(alpha * n)])
        glow = int(d[-int(alpha * n)])
        xaxis[c] = np.linspace(-50, gmax + 50, 400)
        scale = ghigh - glow
        with np.errstate(divide='ignore', invalid='ignore'):
            prior[c] = np.where((xaxis[c] >= glow) & (xaxis[c] <= ghigh), 1.0 / scale, 0.0) if scale > 0 else np.nan
    return xaxis, prior


def density_greedy(densities, labels, xaxis, prior, bandwidth, select_nums):
    """densities (N,D) f32 cuda, labels (N,D) i32 cuda (0 = padding) -> order (select_nums) int32 cuda, scores f64"""
    require_cuda(densities, labels)
    dev = densities.device
    N, D = densities.shape
    C = xaxis.shape[0]
    xa = torch.from_numpy(np.ascontiguousarray(xaxis)).to(dev)
    pr = torch.from_numpy(np.ascontiguousarray(prior)).to(dev)
    order = torch.empty((select_nums,), dtype=torch.int32, device=dev)
    scores = torch.empty((select_nums,), dtype=torch.float64, device=dev)
    wsb = lib.crb_density_greedy_workspace_bytes(N, C)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    check(lib.crb_density_greedy(ptr(densities.contiguous().float()), ptr(labels.contiguous().int()), N, D, C, ptr(xa),
                                 ptr(pr), float(bandwidth), int(select_nums), ptr(order), ptr(scores), ptr(ws), wsb,
                                 cur_stream(dev)), 'crb_density_greedy')
    return order, scores


def kmeans_plusplus_device(X, n_clusters, random_state=0):
    """k-means++ seeding (the reference's `sklearn.cluster.kmeans_plusplus(X, n_clusters, random_state=0)`,
    crb_sampling.py:225-229) with the embeddings left on the device.

    Restates sklearn 1.7's `_kmeans_plusplus` for float32 input step by step: the first centre and every trial position come
    from the SAME numpy RandomState stream (drawn on the host: one `choice`, then `2 + int(log k)` uniforms per centre);
    squared distances as sklearn's float32 path forms them (`_euclidean_distances_upcast`: -2 x.y + |x|^2 + |y|^2 in
    float64, rounded to float32, clamped at 0); trial positions by searchsorted on the float64 cumulative sum; the trial with
    the smallest float32 potential wins. The 500 x 65536 matrix never crosses PCIe and is not re-converted to float64 for
    each of the 300 centres (10.5 s on the host for K1*N = 500, K2*N = 300).
    Same picks as sklearn unless two trial potentials of one step tie within float32 summation noise (sums are taken in
    another order) — selected with ACTIVE_TRAIN.ACTIVE_CONFIG.CLUSTERING = 'kmeans++_device'; 'kmeans++' keeps sklearn itself.
    -> LongTensor (n_clusters,) of row indices, on X's device"""
    assert X.dim() == 2 and X.dtype == torch.float32
    n = X.shape[0]
    if n < n_clusters:
        raise ValueError(f'n_samples={n} should be >= n_clusters={n_clusters}.')
    rs = np.random.RandomState(random_state) if not isinstance(random_state, np.random.RandomState) else random_state
    n_local = 2 + int(np.log(n_clusters))
    w = np.ones(n, dtype=np.float32)
    first = int(rs.choice(n, p=w / w.sum()))
    U = torch.from_numpy(rs.uniform(size=(max(n_clusters - 1, 0), n_local))).to(X.device)        # float64
    X64 = X.double()
    norms = (X64 * X64).sum(1)

    # x.y for a handful of rows against all rows, K = 65536 deep: as one float64 GEMM it gets a few workgroups (10.5 ms per
    # step on MI355X); cut into S slices of the K axis it is a batched GEMM + a sum over slices (0.09 ms)
    dim = X.shape[1]
    S = 64
    while S > 1 and (dim % S or dim // S < 256):
        S //= 2
    Bs = X64.view(n, S, dim // S).permute(1, 2, 0)                         # (S, k, n) view

    def sq_dist(rows):                                    # (t,) indices -> (t, n) float32
        A = X64[rows].view(-1, S, dim // S).permute(1, 0, 2)               # (S, t, k)
        d = -2.0 * torch.bmm(A, Bs).sum(0)
        d += norms[rows][:, None]
        d += norms[None, :]
        return d.float().clamp_(min=0)

    picks = torch.empty((n_clusters,), dtype=torch.long, device=X.device)
    picks[0] = first
    closest = sq_dist(picks[:1])[0]
    pot = closest.sum()
    for c in range(1, n_clusters):
        rand_vals = U[c - 1] * pot.double()
        cand = torch.searchsorted(torch.cumsum(closest.double(), 0), rand_vals).clamp_(max=n - 1)
        dc = torch.minimum(closest[None, :], sq_dist(cand))
        pots = dc.sum(1)
        best = torch.argmin(pots)
        pot, closest = pots[best], dc[best]
        picks[c] = cand[best]
    return picks
