"""Host side of the HIP sparse-convolution path (C-ABI: crb_sparse_*, crb_subm_*, crb_spconv_*, crb_table*_)."""
import ctypes

import torch

from ._lib import lib, check, ptr, cur_stream, require_cuda, host_i32x3, CrbHipError


def _triple(v):
    if isinstance(v, (list, tuple)):
        assert len(v) == 3
        return [int(x) for x in v]
    return [int(v)] * 3


def conv_out_shape(shape, ksize, stride, padding):
    return [(int(s) + 2 * p - k) // st + 1 for s, k, st, p in zip(shape, ksize, stride, padding)]


CHUNK = lib.crb_table_chunk_rows()       # rows per sort chunk of the kernel order

TILE_LPT = True       # 64-row tiles dispatched heaviest first inside each XCD range (tile_order of crb_tables_finish)
MASK_SORT = True      # set False to run the kernels on the natural row order (A/B measurements, order-independence tests)
# rows are sorted by neighbour mask inside chunks of this many consecutive rows: a global sort maximises MFMA skipping
# (0.89 vs 0.80 useful/issued) but scatters each tile's gathers over the whole feature map (L2 misses); chunks keep the
# spatial locality of the row order. 4096 = the fused path (crb_tables_finish); other values take the one-table-at-a-time
# building blocks (ranked keys + device radix sort) and exist for A/B runs only.
MASK_SORT_CHUNK = int(__import__('os').environ.get('CRB_MASK_SORT_CHUNK', str(CHUNK)))
# output-row-window work decomposition of the wgrad (crb_sparse_conv_wgrad_windowed): L2 hit rate of the gathers 7 % -> 64 %,
# fabric reads 800 -> 350 MB per 64x64 launch, but 150 -> 196 us: the matrix pipe, not the memory side, paces the kernel once
# 3 waves share a SIMD, and 64-96 workgroup slots per XCD cannot be dealt evenly to 27 offsets. Opt-in.
WGRAD_WINDOWED = False
COMPACT_TABLES = True  # mask + packed-index tables for the kernels that have a compact instance (A/B: set False)


class CompactTable(object):
    """what the gather-GEMM reads: per row of the KERNEL order (row perm[i] of the (n,K) table) the mask of present offsets,
    the exclusive prefix of the masks' popcounts and the present neighbour indices packed row after row; order = dispatch
    order of the 64-row tiles (heaviest first inside each XCD range) or None"""
    __slots__ = ('cmask', 'cbase', 'packed', 'perm', 'n', 'K', 'order')

    def __init__(self, cmask, cbase, packed, perm, n, K, order=None):
        self.cmask, self.cbase, self.packed, self.perm, self.n, self.K, self.order = cmask, cbase, packed, perm, n, K, order

    def num_pairs(self):
        return int(self.cbase[-1].item())

    def to_nbr(self):
        """the (n,K) table in kernel order this compact table encodes (tests)"""
        n, K = self.n, self.K
        bits = ((self.cmask.long()[:, None] >> torch.arange(K, device=self.cmask.device)[None, :]) & 1).bool()
        rank = torch.cumsum(bits.long(), 1) - bits.long()
        idx = (self.cbase[:-1].long()[:, None] + rank).clamp(max=max(self.packed.numel() - 1, 0))
        out = torch.where(bits, self.packed.long()[idx], torch.full_like(idx, -1))
        return out.int()


class KernelTable(object):
    """one (n,K) neighbour table of a rulebook with everything derived from it"""
    __slots__ = ('nbr', 'mask', 'hist', 'n', 'K', 'compact', 'pairs', 'legacy')

    def __init__(self, nbr, mask, hist, K):
        self.nbr, self.mask, self.hist, self.n, self.K = nbr, mask, hist, nbr.shape[0], K
        self.compact = self.pairs = self.legacy = None


def _hist_rows(n):
    return max((n + CHUNK - 1) // CHUNK, 1) * 32


def table_from_nbr(nbr, K):
    """KernelTable of an existing (n,K) table: one launch for masks + per-chunk offset counts"""
    dev = nbr.device
    n = nbr.shape[0]
    mask = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
    hist = torch.zeros((_hist_rows(n),), dtype=torch.int32, device=dev)
    check(lib.crb_table_masks(ptr(nbr), n, K, ptr(mask), ptr(hist), cur_stream(dev)), 'crb_table_masks')
    return KernelTable(nbr, mask, hist, K)


class _TablePlan(ctypes.Structure):            # include/crb_hip.h: CrbTablePlan
    _fields_ = [(k, ctypes.c_void_p) for k in ('nbr', 'mask', 'hist', 'perm', 'cmask', 'cbase', 'packed', 'tile_weight',
                                               'tile_order', 'pair_in', 'pair_out', 'pair_start', 'pair_unit_base')] + \
               [('n', ctypes.c_int64), ('K', ctypes.c_int32), ('reserved', ctypes.c_int32)]


def plan_tables(tables, want_pairs):
    """output buffers + the CrbTablePlan array of crb_tables_finish for `tables` -> (plans, buffers)"""
    dev = tables[0].nbr.device
    plans = (_TablePlan * len(tables))()
    keep = []
    # ONE allocation for all outputs of all tables (a hundred torch.empty calls cost more host time than the kernels run)
    need, layout = 0, []
    for t, wp in zip(tables, want_pairs):
        n, K = t.n, t.K
        tiles = (n + 63) // 64
        sizes = [max(n, 1), max(n, 1), n + 1, max(n * K, 1), max(tiles, 1), max(tiles, 1)]
        if wp:
            sizes += [max(n * K, 1), max(n * K, 1), K + 1, max(tiles * 32, 1)]
        offs = []
        for sz in sizes:
            offs.append(need)
            need += (sz + 63) // 64 * 64                       # 256-byte aligned carves
        layout.append((sizes, offs))
    arena = torch.empty((need,), dtype=torch.int32, device=dev)
    for i, (t, wp, (sizes, offs)) in enumerate(zip(tables, want_pairs, layout)):
        bufs = [arena[o:o + sz] for sz, o in zip(sizes, offs)]
        perm, cmask, cbase, packed, weight, order = bufs[:6]
        pin, pout, pstart, ubase = bufs[6:] if wp else (None, None, None, None)
        p = plans[i]
        for name, ten in (('nbr', t.nbr), ('mask', t.mask), ('hist', t.hist), ('perm', perm), ('cmask', cmask), ('cbase', cbase),
                          ('packed', packed), ('tile_weight', weight), ('tile_order', order), ('pair_in', pin),
                          ('pair_out', pout), ('pair_start', pstart), ('pair_unit_base', ubase)):
            setattr(p, name, ten.data_ptr() if ten is not None else None)
        p.n, p.K, p.reserved = t.n, t.K, 0
        keep.append((perm, cmask, cbase, packed, weight, order, pin, pout, pstart))
    return plans, keep


def finish_tables(tables, want_pairs):
    """kernel order, compact table, tile order and (want_pairs[i]) the wgrad pair lists of every KernelTable in `tables`:
    three launches for all of them (crb_tables_finish)"""
    tables = [t for t in tables]
    if not tables:
        return
    dev = tables[0].nbr.device
    plans, keep = plan_tables(tables, want_pairs)
    check(lib.crb_tables_finish(plans, len(tables), cur_stream(dev)), 'crb_tables_finish')
    for t, wp, (perm, cmask, cbase, packed, weight, order, pin, pout, pstart) in zip(tables, want_pairs, keep):
        n = t.n
        t.compact = CompactTable(cmask[:n], cbase, packed, perm[:n] if n else None, n, t.K,
                                 order if (TILE_LPT and n >= 128) else None)
        if wp:
            t.pairs = (pin, pout, pstart)


class Rulebook(object):
    """Indice data of one sparse conv (shared by layers with the same indice_key).

    nbr   (n_out,K) i32 : input row feeding output row through offset o, or -1   (forward table)
    nbr_t (n_in,K)  i32 : output row fed by input row through offset o, or -1    (dgrad table; None for SubM,
                          whose table is its own transpose under o -> K-1-o)
    pairs : (pair_in, pair_out, pair_start) for wgrad
    """

    def __init__(self, nbr, nbr_t, n_in, n_out, ksize, stride, padding, subm, in_shape, out_shape, out_coords,
                 table=None, table_t=None):
        self.nbr, self.nbr_t = nbr, nbr_t
        self.n_in, self.n_out = n_in, n_out
        self.ksize, self.stride, self.padding = ksize, stride, padding
        self.K = ksize[0] * ksize[1] * ksize[2]
        self.subm = subm
        self.in_shape, self.out_shape = in_shape, out_shape
        self.out_coords = out_coords
        self.in_coords = None
        self._tab = {'nbr': table, 'nbr_t': table_t}
        self._alt = {}                      # A/B variants of the kernel tables (natural order, other chunk sizes)

    def table(self, which):
        """KernelTable of 'nbr' | 'nbr_t' (built on first use when the plan did not include it)"""
        t = self._tab[which]
        if t is None:
            t = self._tab[which] = table_from_nbr(self.nbr if which == 'nbr' else self.nbr_t, self.K)
        return t

    def _finished(self, which, pairs=False):
        t = self.table(which)
        if t.compact is None or (pairs and t.pairs is None):
            finish_tables([t], [pairs or t.pairs is not None])
        return t

    def row_perm(self, which):
        """kernel order of the rows of 'nbr' | 'nbr_t' (mask-sorted chunks) or None"""
        return self.compact_table(which).perm

    def compact_table(self, which):
        """('nbr' | 'nbr_t') -> CompactTable"""
        if MASK_SORT and MASK_SORT_CHUNK == CHUNK:
            return self._finished(which).compact
        key = ('compact', which, MASK_SORT, MASK_SORT_CHUNK)
        if key not in self._alt:
            table = self.nbr if which == 'nbr' else self.nbr_t
            self._alt[key] = _compact(table, _mask_perm(table, self.K), self.K)
        return self._alt[key]

    def sorted_table(self, which):
        """('nbr' | 'nbr_t') -> (table rows in kernel order (n,K), perm int32 or None): the layout of the kernels without a
        compact-table instance; the tile order is applied to the rows here (those kernels take no indirection)"""
        key = ('sorted', which, MASK_SORT, MASK_SORT_CHUNK, TILE_LPT)
        if key not in self._alt:
            table = self.nbr if which == 'nbr' else self.nbr_t
            ct = self.compact_table(which)
            perm = ct.perm
            if perm is not None and ct.order is not None:
                full = table.shape[0] // 64
                body = perm[:full * 64].view(full, 64)[ct.order[:full].long()].reshape(-1)
                perm = torch.cat([body, perm[full * 64:]]).contiguous()
            if perm is None:
                self._alt[key] = (table, None)
            else:
                out = torch.empty_like(table)
                check(lib.crb_nbr_permute(ptr(table), ptr(perm), table.shape[0], self.K, ptr(out), cur_stream(table.device)),
                      'crb_nbr_permute')
                self._alt[key] = (out, perm)
        return self._alt[key]

    def table_for(self, which, cin, cout, arithmetic='f32'):
        """the table form the gather-GEMM instance of (cin, cout) consumes"""
        if COMPACT_TABLES and (lib.crb_sparse_conv_compact_supported(cin, cout) or
                               (arithmetic == 'bf16x3' and _bf16x3_supported(cin, cout))):
            return self.compact_table(which)
        return self.sorted_table(which)

    def _pairs_of(self, which):
        p = self._finished(which, pairs=True).pairs
        if WGRAD_WINDOWED and len(p) == 3:
            t = self._tab[which]
            bnd = torch.empty((self.K, lib.crb_wgrad_num_windows() + 1), dtype=torch.int32, device=p[0].device)
            check(lib.crb_wgrad_window_bounds(ptr(p[1]), ptr(p[2]), self.K, t.n, ptr(bnd), cur_stream(bnd.device)),
                  'crb_wgrad_window_bounds')
            p = t.pairs = (p[0], p[1], p[2], bnd)
        return p

    def pairs(self):
        return self._pairs_of('nbr')

    def pairs_t(self):
        """pairs of the transposed conv (inverse conv): 'in' = rows of this conv's output"""
        return self._pairs_of('nbr_t')


def _compact(table, perm, K):
    """one-table-at-a-time compact table (A/B paths: natural order, other chunk sizes)"""
    n = table.shape[0]
    dev = table.device
    cmask = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
    cbase = torch.empty((n + 1,), dtype=torch.int32, device=dev)
    packed = torch.empty((max(n * K, 1),), dtype=torch.int32, device=dev)     # worst case; only cbase[n] entries are touched
    wsb = lib.crb_nbr_compact_workspace_bytes(n)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    check(lib.crb_nbr_compact(ptr(table), ptr(perm), n, K, ptr(cmask), ptr(cbase), ptr(packed), ptr(ws), wsb, cur_stream(dev)),
          'crb_nbr_compact')
    return CompactTable(cmask[:n], cbase, packed, perm, n, K)


def _mask_perm(table, K):
    """one-table-at-a-time row permutation (A/B paths): mask -> chunk sort -> tiles heaviest first, rows physically ordered"""
    n = table.shape[0]
    dev = table.device
    if n == 0 or not MASK_SORT:
        return None
    mask = torch.empty((n,), dtype=torch.int32, device=dev)
    check(lib.crb_nbr_masks(ptr(table), n, K, ptr(mask), cur_stream(dev)), 'crb_nbr_masks')
    if MASK_SORT_CHUNK == lib.crb_mask_sort_chunk_rows():
        perm = torch.empty((n,), dtype=torch.int32, device=dev)            # one LDS bitonic-sort workgroup per chunk
        check(lib.crb_mask_sort_chunks(ptr(mask), n, ptr(perm), cur_stream(dev)), 'crb_mask_sort_chunks')
    elif MASK_SORT_CHUNK > 0 and MASK_SORT_CHUNK % 1024 == 0:
        perm = torch.empty((n,), dtype=torch.int32, device=dev)            # ranked keys per chunk + one device radix sort
        wsb = lib.crb_mask_sort_rows_workspace_bytes(n)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
        check(lib.crb_mask_sort_rows(ptr(mask), n, MASK_SORT_CHUNK, ptr(perm), ptr(ws), wsb, cur_stream(dev)),
              'crb_mask_sort_rows')
    else:
        if MASK_SORT_CHUNK > 0:
            key = (torch.arange(n, device=dev, dtype=torch.int64) // MASK_SORT_CHUNK) * (1 << 32) + \
                (mask.long() & 0xffffffff)
        else:
            key = mask.long() & 0xffffffff
        perm = torch.sort(key, stable=True)[1].to(torch.int32)
    if TILE_LPT and n >= 128:
        wsb = lib.crb_tile_lpt_workspace_bytes(n)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
        perm2 = torch.empty_like(perm)
        check(lib.crb_tile_lpt_perm(ptr(mask), ptr(perm), n, ptr(perm2), ptr(ws), wsb, cur_stream(dev)),
              'crb_tile_lpt_perm')
        perm = perm2
    return perm


def build_hash(coords, shape):
    require_cuda(coords)
    n = coords.shape[0]
    cap = lib.crb_hash_capacity_for(n)
    hkeys = torch.empty((cap,), dtype=torch.int64, device=coords.device)
    hvals = torch.empty((cap,), dtype=torch.int32, device=coords.device)
    check(lib.crb_sparse_hash_build(ptr(coords), n, host_i32x3(shape), ptr(hkeys), ptr(hvals), cap,
                                    cur_stream(coords.device)), 'crb_sparse_hash_build')
    return hkeys, hvals, cap


def _host_i32(vals):
    return (ctypes.c_int32 * len(vals))(*[int(v) for v in vals])


def _host_i64(vals):
    return (ctypes.c_int64 * len(vals))(*[int(v) for v in vals])


class PendingChain(object):
    """First half of build_rulebooks(..., n_dev=...) issued AHEAD of time (begin_rulebooks): the output sets of the strided convs are
    marked and counted on the device and the counts (+ the voxel count) travel to pinned host memory behind an event. The second
    half (build_rulebooks(pending=...)) waits for the event - long past when the marks were enqueued before a whole backward pass -
    instead of stalling the host, and with it the launch queue, in the middle of the forward pass."""
    __slots__ = ('coords', 'shape', 'batch_size', 'specs', 'state', 'host', 'event')

    def counts(self):
        self.event.synchronize()
        return self.host.tolist()


def _chain_mark(coords, shape, batch_size, strided, n_dev, st):
    """bitmaps + per-level counts of the chain of strided convs (device only, no read-back) -> state dict"""
    dev = coords.device
    L = len(strided)
    if L > 8:
        raise CrbHipError('more than 8 strided convs in one chain')
    geoms, oshapes, cur = [], [], list(shape)
    for _, ks, sd, pd in strided:
        o = conv_out_shape(cur, ks, sd, pd)
        if min(o) <= 0:
            raise CrbHipError(f'sparse conv output shape {o} is empty')
        geoms += ks + sd + pd
        oshapes.append(o)
        cur = o
    flat_shapes = _host_i32([v for o in oshapes for v in o])
    word_off = [0]
    for o in oshapes:
        word_off.append(word_off[-1] + lib.crb_spconv_padded_words(batch_size, host_i32x3(o)))
    woff = _host_i64(word_off)
    bitmap_all = torch.empty((word_off[-1],), dtype=torch.int32, device=dev)
    tile_sums = torch.empty((word_off[-1] // 2048,), dtype=torch.int32, device=dev)
    counts = torch.empty((L,), dtype=torch.int32, device=dev)
    if n_dev is not None:
        check(lib.crb_spconv_chain_mark_lazy(ptr(coords), coords.shape[0], ptr(n_dev), batch_size, host_i32x3(shape), L,
                                             _host_i32(geoms), flat_shapes, woff, ptr(bitmap_all), ptr(tile_sums), ptr(counts), st),
              'crb_spconv_chain_mark_lazy')
    else:
        check(lib.crb_spconv_chain_mark(ptr(coords), coords.shape[0], batch_size, host_i32x3(shape), L, _host_i32(geoms),
                                        flat_shapes, woff, ptr(bitmap_all), ptr(tile_sums), ptr(counts), st),
              'crb_spconv_chain_mark')
    return dict(L=L, oshapes=oshapes, flat_shapes=flat_shapes, word_off=word_off, woff=woff, bitmap_all=bitmap_all,
                tile_sums=tile_sums, counts=counts)


def begin_rulebooks(coords, shape, batch_size, specs, n_dev):
    """build_rulebooks' device-only first half for a lazily voxelized coordinate set (n_dev (1,) i32 cuda), issued on the current
    stream with NO host synchronisation -> PendingChain, to be handed to build_rulebooks(pending=...) later (same stream)."""
    require_cuda(coords)
    assert coords.dtype == torch.int32 and coords.is_contiguous() and n_dev is not None
    nspecs = [(s[0], _triple(s[1])) + tuple(_triple(v) for v in s[2:]) for s in specs]
    strided = [s for s in nspecs if s[0] == 'spconv']
    if not strided:
        raise CrbHipError('begin_rulebooks needs at least one strided conv in the chain')
    p = PendingChain()
    p.coords, p.shape, p.batch_size, p.specs = coords, list(shape), batch_size, list(specs)
    p.state = _chain_mark(coords, shape, batch_size, strided, n_dev, cur_stream(coords.device))
    both = torch.cat([n_dev.view(1), p.state['counts']])
    p.host = torch.empty((both.shape[0],), dtype=torch.int32, pin_memory=True)
    p.host.copy_(both, non_blocking=True)
    p.event = torch.cuda.Event()
    p.event.record()
    return p


def build_rulebooks(coords, shape, batch_size, specs, want_grad=True, n_dev=None, pending=None):
    """Rulebooks of a CHAIN of sparse convs over one coordinate set, every table finished, in ~3 launches per table and ONE
    host read-back for the whole chain.

    specs, in forward order: ('subm', ksize) or ('spconv', ksize, stride, padding); a strided conv consumes the current set
    and its output set becomes the current one. want_grad: also the transposed tables of the strided convs (dgrad) and the
    wgrad pair lists; without it (inference) they are built on first use.
    coords (N,4) i32 cuda contiguous [b,z,y,x], unique rows in any order -> list of Rulebook (one per spec).
    n_dev (1,) i32 cuda: the row count is still on the device (crbhip.voxel.voxelize(lazy=True): coords is the generator's
    capacity buffer) - the chain is marked before the host knows N and the chain's read-back returns N as well; -> (books, N)
    (the caller slices its tensors to N). Needs at least one strided conv in `specs`."""
    require_cuda(coords)
    assert coords.dtype == torch.int32 and coords.is_contiguous()
    dev = coords.device
    lazy = n_dev is not None or pending is not None
    if pending is not None:
        # pending: the PendingChain of begin_rulebooks() for exactly this coordinate buffer and chain - its marks and counts are used
        # instead of a new marking pass + read-back (-> (books, N) like the n_dev form)
        assert pending.coords.data_ptr() == coords.data_ptr() and list(pending.shape) == list(shape) and \
            pending.batch_size == batch_size and len(pending.specs) == len(specs)
    if lazy and not any(s[0] == 'spconv' for s in specs):
        n_host = int(n_dev.cpu()[0])                     # nothing to merge the read-back with
        return build_rulebooks(coords[:n_host], shape, batch_size, specs, want_grad), n_host
    st = cur_stream(dev)
    specs = [(s[0], _triple(s[1])) + tuple(_triple(v) for v in s[2:]) for s in specs]
    strided = [s for s in specs if s[0] == 'spconv']
    # ---- the chain of output sets: bitmaps -> counts (the read-back) -> coordinates + rank tables
    levels = []                                   # per strided conv: (out_shape, n_out, out_coords, rank pointer)
    if strided:
        if pending is not None:
            stt = pending.state
            both = pending.counts()
            coords = coords[:int(both[0])]
            n_outs = [int(v) for v in both[1:]]
        else:
            stt = _chain_mark(coords, shape, batch_size, strided, n_dev, st)
            if lazy:
                both = torch.cat([n_dev.view(1), stt['counts']]).cpu().tolist()    # voxel count + level sizes: ONE read-back
                coords = coords[:int(both[0])]
                n_outs = [int(v) for v in both[1:]]
            else:
                n_outs = [int(v) for v in stt['counts'].cpu().tolist()]            # the single read-back of the chain
        L, oshapes, flat_shapes, word_off, woff = stt['L'], stt['oshapes'], stt['flat_shapes'], stt['word_off'], stt['woff']
        bitmap_all, tile_sums = stt['bitmap_all'], stt['tile_sums']
        rank_all = torch.empty((word_off[-1], 2), dtype=torch.int32, device=dev)
        out_coords = [torch.empty((max(n, 1), 4), dtype=torch.int32, device=dev) for n in n_outs]
        oc_ptrs = (ctypes.c_void_p * L)(*[c.data_ptr() for c in out_coords])
        check(lib.crb_spconv_chain_emit(batch_size, L, flat_shapes, woff, ptr(bitmap_all), ptr(tile_sums), ptr(rank_all),
                                        oc_ptrs, _host_i64(n_outs), st), 'crb_spconv_chain_emit')
        for l in range(L):
            levels.append((oshapes[l], n_outs[l], out_coords[l][:n_outs[l]],
                           ctypes.c_void_p(rank_all.data_ptr() + word_off[l] * 8)))
    # ---- neighbour rows: one kernel per table; all per-chunk offset counts live in one zero-filled buffer
    cur_coords, cur_shape, cur_rank, cur_n = coords, list(shape), None, coords.shape[0]
    sizes, lvl = [], 0
    for s in specs:
        K = s[1][0] * s[1][1] * s[1][2]
        if K > 32:
            raise CrbHipError('kernel volumes above 32 offsets have no gfx950 table kernels')
        if s[0] == 'subm':
            sizes.append((cur_n,))
        else:
            sizes.append((levels[lvl][1], cur_n))
            cur_n = levels[lvl][1]
            lvl += 1
    hist_all = torch.zeros((sum(_hist_rows(n) for t in sizes for n in t),), dtype=torch.int32, device=dev)
    hoff = [0]

    def take_hist(n):
        h = hist_all[hoff[0]:hoff[0] + _hist_rows(n)]
        hoff[0] += _hist_rows(n)
        return h

    def i32(*shape_):
        return torch.empty(shape_, dtype=torch.int32, device=dev)
    cur_n, lvl, site_hash, books, to_finish, pairs_flag = coords.shape[0], 0, None, [], [], []
    for s in specs:
        ks = s[1]
        K = ks[0] * ks[1] * ks[2]
        if s[0] == 'subm':
            n = cur_n
            nbr, mask, hist = i32(n, K), i32(max(n, 1)), take_hist(n)
            if cur_rank is None and site_hash is None and n > 0:
                site_hash = build_hash(cur_coords, cur_shape)
            hk, hv, cap = site_hash if cur_rank is None and site_hash is not None else (None, None, 0)
            check(lib.crb_subm_rows(ptr(cur_coords), n, host_i32x3(cur_shape), host_i32x3(ks), ptr(hk), ptr(hv), cap, cur_rank,
                                    ptr(nbr), ptr(mask), ptr(hist), st), 'crb_subm_rows')
            tab = KernelTable(nbr, mask, hist, K)
            rb = Rulebook(nbr, None, n, n, ks, [1, 1, 1], [k // 2 for k in ks], True, list(cur_shape), list(cur_shape),
                          cur_coords, tab, None)
            rb.in_coords = cur_coords
            to_finish.append(tab)
            pairs_flag.append(want_grad)
        else:
            _, _, sd, pd = s
            oshape, n_out, ocoords, orank = levels[lvl]
            lvl += 1
            n = cur_n
            nbr, nbr_t = i32(n_out, K), i32(n, K)
            mask_t, hist_t = i32(max(n, 1)), take_hist(n)
            check(lib.crb_spconv_rows(ptr(cur_coords), n, host_i32x3(ks), host_i32x3(sd), host_i32x3(pd), host_i32x3(oshape),
                                      orank, n_out, ptr(nbr), ptr(nbr_t), ptr(mask_t), ptr(hist_t), st), 'crb_spconv_rows')
            mask, hist = i32(max(n_out, 1)), take_hist(n_out)
            check(lib.crb_table_masks(ptr(nbr), n_out, K, ptr(mask), ptr(hist), st), 'crb_table_masks')
            tab, tab_t = KernelTable(nbr, mask, hist, K), KernelTable(nbr_t, mask_t, hist_t, K)
            rb = Rulebook(nbr, nbr_t, n, n_out, ks, sd, pd, False, list(cur_shape), list(oshape), ocoords, tab, tab_t)
            rb.in_coords = cur_coords
            to_finish.append(tab)
            pairs_flag.append(want_grad)
            if want_grad:
                to_finish.append(tab_t)
                pairs_flag.append(False)
            cur_coords, cur_shape, cur_rank, cur_n, site_hash = ocoords, list(oshape), orank, n_out, None
        books.append(rb)
    if MASK_SORT and MASK_SORT_CHUNK == CHUNK:
        finish_tables(to_finish, pairs_flag)
    return (books, coords.shape[0]) if lazy else books


def subm_rulebook(coords, shape, ksize):
    """coords (N,4) i32 cuda contiguous [b,z,y,x]"""
    return build_rulebooks(coords, shape, 1, [('subm', ksize)], want_grad=torch.is_grad_enabled())[0]


def spconv_rulebook(coords, shape, batch_size, ksize, stride, padding):
    return build_rulebooks(coords, shape, batch_size, [('spconv', ksize, stride, padding)],
                           want_grad=torch.is_grad_enabled())[0]


# When set to a list, every gather-GEMM launch appends (kind, cin, cout, K, n_in, n_out, nbr, ev0, ev1): HIP events on
# the launch stream (torch's current stream IS the stream handed to the C-ABI), read back by bench.py for the roofline.
PROFILE = None
# Arithmetic contract of the gather-GEMM (forward and dgrad) is an ARGUMENT of the call (sparse_conv(..., arithmetic=...);
# spconv.pytorch.SparseConvolution.arithmetic on the module side), never process state. 'f32' (default): exact f32 MFMA.
# 'bf16x3' (OPT-IN): operands split into two bf16 values, three bf16 MFMA passes, f32 accumulation:
# |y - y_f32| <= 2^-16 sum |x||w| (include/crb_hip_measure.h, crb_sparse_conv_forward_bf16x3); shapes without a bf16x3 instance
# (C <= 16) keep the f32 kernel. Since round 5 the bf16x3 kernel lives in the MEASUREMENT library only (it buys nothing on the default
# path: VERDICT r04 item 8): asking for it in a process that loaded the product library raises CrbHipError.
ARITHMETICS = ('f32', 'bf16x3')


def _bf16x3_supported(cin, cout):
    from ._lib import require_measure
    require_measure('crb_sparse_conv_forward_bf16x3')
    return bool(lib.crb_sparse_conv_bf16x3_supported(cin, cout))


def epilogue_supported(cin, cout):
    """shapes whose forward kernel can apply (bias,) BatchNorm1d(eval) and ReLU to the accumulator before the store"""
    return bool(COMPACT_TABLES and lib.crb_sparse_conv_compact_supported(cin, cout))


def _conv_forward_raw(x, w_kio, table, n_out, kind='fwd', epilogue=None, arithmetic='f32'):
    """x (n_in,cin), w (K,cin,cout), table = (nbr rows in kernel order (n_out,K), perm or None) or a CompactTable
    -> (n_out,cout). epilogue = (bias or None, BatchNorm1d in eval mode, relu flag): inference only, compact tables only."""
    K, cin, cout = w_kio.shape
    if not lib.crb_sparse_conv_supported(cin, cout):
        raise CrbHipError(f'sparse conv channel pair ({cin},{cout}) has no gfx950 kernel instance')
    y = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
    prof = PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    if epilogue is not None:
        bias, bn, relu = epilogue
        assert isinstance(table, CompactTable) and not bn.training
        check(lib.crb_sparse_conv_forward_compact_bn(
            ptr(x), ptr(w_kio), ptr(table.cmask), ptr(table.cbase), ptr(table.packed), ptr(table.perm), ptr(table.order), ptr(y), n_out,
            K, cin, cout, ptr(bias.contiguous()) if bias is not None else None, ptr(bn.weight.contiguous()), ptr(bn.bias.contiguous()),
            ptr(bn.running_mean.contiguous()), ptr(bn.running_var.contiguous()), float(bn.eps), int(bool(relu)),
            cur_stream(x.device)), 'crb_sparse_conv_forward_compact_bn')
        nbr = table
        kind = kind + '_bn'
    elif arithmetic == 'bf16x3' and isinstance(table, CompactTable) and _bf16x3_supported(cin, cout):
        wsb = lib.crb_sparse_conv_bf16x3_workspace_bytes(K, cin, cout)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=x.device)
        check(lib.crb_sparse_conv_forward_bf16x3(ptr(x), ptr(w_kio), ptr(table.cmask), ptr(table.cbase), ptr(table.packed),
                                                 ptr(table.perm), ptr(table.order), ptr(y), x.shape[0], n_out, K, cin, cout, ptr(ws), wsb,
                                                 cur_stream(x.device)), 'crb_sparse_conv_forward_bf16x3')
        nbr = table
        kind = kind + '_bf16x3'
    elif isinstance(table, CompactTable):
        check(lib.crb_sparse_conv_forward_compact(ptr(x), ptr(w_kio), ptr(table.cmask), ptr(table.cbase), ptr(table.packed),
                                                  ptr(table.perm), ptr(table.order), ptr(y), n_out, K, cin, cout,
                                                  cur_stream(x.device)), 'crb_sparse_conv_forward_compact')
        nbr = table
    else:
        nbr, perm = table
        check(lib.crb_sparse_conv_forward(ptr(x), ptr(w_kio), ptr(nbr), ptr(perm), ptr(y), n_out, K, cin, cout,
                                          cur_stream(x.device)),
              'crb_sparse_conv_forward')
    if prof is not None:
        ev1.record()
        prof.append((kind, cin, cout, K, x.shape[0], n_out, nbr, ev0, ev1))
    return y


def _conv_wgrad_raw(x, dy, pairs, K, kind='wgrad'):
    cin, cout = x.shape[1], dy.shape[1]
    pin, pout, pstart = pairs[:3]
    bnd = pairs[3] if len(pairs) > 3 else None
    dw = torch.empty((K, cin, cout), dtype=torch.float32, device=x.device)
    windowed = WGRAD_WINDOWED and bnd is not None and lib.crb_sparse_conv_wgrad_windowed_supported(cin, cout)
    wsb = lib.crb_sparse_conv_wgrad_windowed_workspace_bytes(K, cin, cout) if windowed else \
        lib.crb_sparse_conv_wgrad_workspace_bytes(K, cin, cout)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=x.device)
    prof = PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    if windowed:
        check(lib.crb_sparse_conv_wgrad_windowed(ptr(x), ptr(dy), ptr(pin), ptr(pout), ptr(pstart), ptr(bnd), ptr(dw), K, cin,
                                                 cout, ptr(ws), wsb, cur_stream(x.device)), 'crb_sparse_conv_wgrad_windowed')
    else:
        check(lib.crb_sparse_conv_wgrad(ptr(x), ptr(dy), ptr(pin), ptr(pout), ptr(pstart), ptr(dw), K, cin, cout, ptr(ws),
                                        wsb, cur_stream(x.device)), 'crb_sparse_conv_wgrad')
    if prof is not None:
        ev1.record()
        prof.append((kind, cin, cout, K, x.shape[0], dy.shape[0], pstart, ev0, ev1))
    return dw


# weight layouts made ahead for all layers of a step (crb_sparse_weights_multi): parameter key -> (w_kio, w_dgrad, flip), and the same
# record under w_kio's address for the backward. Keys carry the parameter's autograd version: no stale layout is handed out.
_PREP_W = {}
_PREP_WD = {}
PREPARE_WEIGHTS = __import__('os').environ.get('CRB_SPARSE_PREPARE', '1') == '1'


def _wkey(w):
    return (w.data_ptr(), w._version, tuple(w.shape), w.device.index)


def prepare_weights(convs):
    """convs: spconv layers (weight (Cout, k.., Cin) contiguous f32, .subm) -> their (K,Cin,Cout) forward operands and (K,Cout,Cin)
    input-gradient operands in ONE launch instead of three launch-bound launches per layer; weight_kio / SparseConvFunction.backward
    pick them up"""
    import ctypes
    _PREP_W.clear()
    _PREP_WD.clear()
    jobs = [c for c in convs if PREPARE_WEIGHTS and c.weight.is_cuda and c.weight.dtype == torch.float32 and c.weight.is_contiguous()]
    for lo in range(0, len(jobs), 32):
        part = jobs[lo:lo + 32]
        n = len(part)
        dims = []
        for c in part:
            w = c.weight
            cout, cin = int(w.shape[0]), int(w.shape[-1])
            dims.append((w.numel() // (cout * cin), cin, cout))
        kio = [torch.empty(d, dtype=torch.float32, device=c.weight.device) for c, d in zip(part, dims)]
        wd = [torch.empty((d[0], d[2], d[1]), dtype=torch.float32, device=c.weight.device) for c, d in zip(part, dims)]
        A = lambda t, v: (t * n)(*v)
        check(lib.crb_sparse_weights_multi(n, A(ctypes.c_void_p, [c.weight.data_ptr() for c in part]), A(ctypes.c_int32, [d[0] for d in dims]),
                                           A(ctypes.c_int32, [d[1] for d in dims]), A(ctypes.c_int32, [d[2] for d in dims]),
                                           A(ctypes.c_int32, [int(bool(c.subm)) for c in part]),
                                           A(ctypes.c_void_p, [t.data_ptr() for t in kio]), A(ctypes.c_void_p, [t.data_ptr() for t in wd]),
                                           cur_stream(part[0].weight.device)), 'crb_sparse_weights_multi')
        for c, a, b in zip(part, kio, wd):
            # (the record keeps the parameter's storage alive: while it exists no other tensor can sit at that address, so address +
            #  version + shape name THIS parameter's values and nothing else)
            rec = (a, b, bool(c.subm), c.weight.detach())
            _PREP_W[_wkey(c.weight)] = rec
            _PREP_WD[a.data_ptr()] = rec
    return len(jobs)


class _WeightKIO(torch.autograd.Function):
    """(Cout, k.., Cin) parameter -> (K, Cin, Cout) contiguous forward operand; the prepared copy when there is one"""

    @staticmethod
    def forward(ctx, weight, K):
        ctx.wshape = weight.shape
        hit = _PREP_W.get(_wkey(weight)) if _PREP_W else None
        if hit is not None:
            return hit[0]
        return weight.reshape(weight.shape[0], K, weight.shape[-1]).permute(1, 2, 0).contiguous()

    @staticmethod
    def backward(ctx, g):
        return g.permute(2, 0, 1).reshape(ctx.wshape), None


def weight_kio(weight, K):
    return _WeightKIO.apply(weight, K)


def _dgrad_weights(w, subm):
    """W[o]^T (at K-1-o for submanifold layers) of a (K,Cin,Cout) forward operand: prepared or formed here"""
    hit = _PREP_WD.get(w.data_ptr()) if _PREP_WD else None
    if hit is not None and hit[0].data_ptr() == w.data_ptr() and hit[0].shape == w.shape and hit[2] == bool(subm):
        return hit[1]
    return (w.flip(0) if subm else w).transpose(1, 2).contiguous()


class SparseConvFunction(torch.autograd.Function):
    """y = sum_o x[nbr[:,o]] @ w[o]   (w in (K,Cin,Cout) layout; `inverse` runs the transposed rulebook)"""

    @staticmethod
    def forward(ctx, x, w_kio, rb, inverse, arithmetic='f32'):
        require_cuda(x, w_kio)
        if arithmetic not in ARITHMETICS:
            raise CrbHipError('unknown gather-GEMM arithmetic %r' % (arithmetic,))
        x = x.contiguous().float()
        w_kio = w_kio.contiguous().float()
        K, cin, cout = w_kio.shape
        if inverse:
            table, n_out = rb.table_for('nbr_t', cin, cout, arithmetic), rb.n_in
        else:
            table, n_out = rb.table_for('nbr', cin, cout, arithmetic), rb.n_out
        ctx.rb, ctx.inverse, ctx.arithmetic = rb, inverse, arithmetic
        ctx.save_for_backward(x, w_kio)
        return _conv_forward_raw(x, w_kio, table, n_out, ('subm' if rb.subm else 'spconv') + '_fwd', arithmetic=arithmetic)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        rb, inverse, ar = ctx.rb, ctx.inverse, ctx.arithmetic
        dy = dy.contiguous().float()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if rb.subm:
                wd = _dgrad_weights(w, True)                     # Wd[o] = W[K-1-o]^T
                dx = _conv_forward_raw(dy, wd, rb.table_for('nbr', wd.shape[1], wd.shape[2], ar), rb.n_in, 'subm_dgrad',
                                       arithmetic=ar)
            else:
                wd = _dgrad_weights(w, False)
                table, n_in = (rb.table_for('nbr', wd.shape[1], wd.shape[2], ar), rb.n_out) if inverse else \
                    (rb.table_for('nbr_t', wd.shape[1], wd.shape[2], ar), rb.n_in)
                dx = _conv_forward_raw(dy, wd, table, n_in, 'spconv_dgrad', arithmetic=ar)
        if ctx.needs_input_grad[1]:
            pairs = rb.pairs_t() if inverse else rb.pairs()
            dw = _conv_wgrad_raw(x, dy, pairs, rb.K, ('subm' if rb.subm else 'spconv') + '_wgrad')
        return dx, dw, None, None, None


def sparse_conv(x, w_kio, rb, inverse=False, arithmetic='f32'):
    return SparseConvFunction.apply(x, w_kio, rb, inverse, arithmetic)


@torch.no_grad()
def sparse_conv_bn_eval(x, w_kio, rb, bias, bn, relu):
    """inference: conv (+bias) -> BatchNorm1d(running statistics) -> ReLU in ONE launch (epilogue on the accumulator)"""
    require_cuda(x, w_kio)
    x, w_kio = x.contiguous().float(), w_kio.contiguous().float()
    cin, cout = w_kio.shape[1], w_kio.shape[2]
    return _conv_forward_raw(x, w_kio, rb.table_for('nbr', cin, cout), rb.n_out,
                             ('subm' if rb.subm else 'spconv') + '_fwd', epilogue=(bias, bn, relu))


class ToDenseFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, coords, batch_size, shape):
        require_cuda(feat, coords)
        feat = feat.contiguous().float()
        n, C = feat.shape
        D, H, W = [int(v) for v in shape]
        out = torch.empty((batch_size, C, D, H, W), dtype=torch.float32, device=feat.device)
        check(lib.crb_sparse_to_dense(ptr(feat), ptr(coords), ptr(out), n, batch_size, C, D, H, W, 1,
                                      cur_stream(feat.device)), 'crb_sparse_to_dense')
        ctx.save_for_backward(coords)
        ctx.meta = (n, batch_size, C, D, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        (coords,) = ctx.saved_tensors
        n, B, C, D, H, W = ctx.meta
        g = g.contiguous().float()
        df = torch.empty((n, C), dtype=torch.float32, device=g.device)
        check(lib.crb_dense_to_sparse(ptr(g), ptr(coords), ptr(df), n, B, C, D, H, W, cur_stream(g.device)),
              'crb_dense_to_sparse')
        return df, None, None, None


def to_dense(feat, coords, batch_size, shape):
    return ToDenseFunction.apply(feat, coords, batch_size, shape)


class ToBEVChannelsLastFunction(torch.autograd.Function):
    """HeightCompression in one step: (N,C) sparse rows -> (B, C*D, H, W) tensor in channels_last memory format"""

    @staticmethod
    def forward(ctx, feat, coords, batch_size, shape):
        require_cuda(feat, coords)
        feat = feat.contiguous().float()
        n, C = feat.shape
        D, H, W = [int(v) for v in shape]
        out = torch.empty((batch_size, C * D, H, W), dtype=torch.float32, device=feat.device,
                          memory_format=torch.channels_last)
        check(lib.crb_sparse_to_dense_nhwc(ptr(feat), ptr(coords), ctypes_ptr(out), n, batch_size, C, D, H, W, 1,
                                           cur_stream(feat.device)), 'crb_sparse_to_dense_nhwc')
        ctx.save_for_backward(coords)
        ctx.meta = (n, batch_size, C, D, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        (coords,) = ctx.saved_tensors
        n, B, C, D, H, W = ctx.meta
        g = g.float().contiguous(memory_format=torch.channels_last)
        df = torch.empty((n, C), dtype=torch.float32, device=g.device)
        check(lib.crb_dense_to_sparse_nhwc(ctypes_ptr(g), ptr(coords), ptr(df), n, B, C, D, H, W, cur_stream(g.device)),
              'crb_dense_to_sparse_nhwc')
        return df, None, None, None


def ctypes_ptr(t):
    """data pointer of a tensor that is dense in SOME memory format (channels_last tensors are not .is_contiguous())"""
    import ctypes
    return ctypes.c_void_p(t.data_ptr())


def to_bev_channels_last(feat, coords, batch_size, shape):
    return ToBEVChannelsLastFunction.apply(feat, coords, batch_size, shape)
