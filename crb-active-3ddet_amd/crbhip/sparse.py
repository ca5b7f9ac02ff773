"""Host side of the HIP sparse-convolution path (C-ABI: crb_sparse_*, crb_subm_*, crb_spconv_*, crb_pairs_*)."""
import torch

from ._lib import lib, check, ptr, cur_stream, require_cuda, host_i32x3, CrbHipError


def _triple(v):
    if isinstance(v, (list, tuple)):
        assert len(v) == 3
        return [int(x) for x in v]
    return [int(v)] * 3


def conv_out_shape(shape, ksize, stride, padding):
    return [(int(s) + 2 * p - k) // st + 1 for s, k, st, p in zip(shape, ksize, stride, padding)]


class Rulebook(object):
    """Indice data of one sparse conv (shared by layers with the same indice_key).

    nbr   (n_out,K) i32 : input row feeding output row through offset o, or -1   (forward table)
    nbr_t (n_in,K)  i32 : output row fed by input row through offset o, or -1    (dgrad table; None for SubM,
                          whose table is its own transpose under o -> K-1-o)
    pairs : (pair_in, pair_out, pair_start) built lazily for wgrad
    """

    def __init__(self, nbr, nbr_t, n_in, n_out, ksize, stride, padding, subm, in_shape, out_shape, out_coords):
        self.nbr, self.nbr_t = nbr, nbr_t
        self.n_in, self.n_out = n_in, n_out
        self.ksize, self.stride, self.padding = ksize, stride, padding
        self.K = ksize[0] * ksize[1] * ksize[2]
        self.subm = subm
        self.in_shape, self.out_shape = in_shape, out_shape
        self.out_coords = out_coords
        self.in_coords = None
        self._pairs = None
        self._pairs_t = None
        self._sorted = {}

    def row_perm(self, which):
        """kernel order of the rows of 'nbr' | 'nbr_t' (mask-sorted chunks, tiles heaviest first) or None"""
        key = 'perm_' + which
        if key not in self._sorted:
            table = self.nbr if which == 'nbr' else self.nbr_t
            self._sorted[key] = _mask_perm(table, self.K)
        return self._sorted[key]

    def sorted_table(self, which):
        """('nbr' | 'nbr_t') -> (table rows in neighbour-mask order, perm int32): the (n,K) layout of the kernels without a
        compact-table instance (rows of a wave share their set of active kernel offsets)"""
        if which not in self._sorted:
            table = self.nbr if which == 'nbr' else self.nbr_t
            perm = self.row_perm(which)
            if perm is None:
                self._sorted[which] = (table, None)
            else:
                out = torch.empty_like(table)
                check(lib.crb_nbr_permute(ptr(table), ptr(perm), table.shape[0], self.K, ptr(out), cur_stream(table.device)),
                      'crb_nbr_permute')
                self._sorted[which] = (out, perm)
        return self._sorted[which]

    def compact_table(self, which):
        """('nbr' | 'nbr_t') -> CompactTable: per row of the kernel order a mask of present offsets + the present
        neighbour indices packed row after row (crb_nbr_compact): 8 + 4 P/n bytes per row instead of 4 K"""
        key = 'compact_' + which
        if key not in self._sorted:
            table = self.nbr if which == 'nbr' else self.nbr_t
            self._sorted[key] = _compact(table, self.row_perm(which), self.K)
        return self._sorted[key]

    def table_for(self, which, cin, cout, arithmetic='f32'):
        """the table form the gather-GEMM instance of (cin, cout) consumes"""
        if COMPACT_TABLES and (lib.crb_sparse_conv_compact_supported(cin, cout) or
                               (arithmetic == 'bf16x3' and lib.crb_sparse_conv_bf16x3_supported(cin, cout))):
            return self.compact_table(which)
        return self.sorted_table(which)

    def pairs(self):
        if self._pairs is None:
            self._pairs = _pairs_from_nbr(self.nbr, self.n_out, self.K)
        return self._pairs

    def pairs_t(self):
        """pairs of the transposed conv (inverse conv): 'in' = rows of this conv's output"""
        if self._pairs_t is None:
            self._pairs_t = _pairs_from_nbr(self.nbr_t, self.n_in, self.K)
        return self._pairs_t


TILE_LPT = True       # 64-row tiles of the sorted table dispatched heaviest first (crb_tile_lpt_perm)
MASK_SORT = True      # set False to run the kernel on the natural row order (A/B measurements)
# rows are sorted by neighbour mask inside chunks of this many consecutive rows: a global sort maximises MFMA skipping
# (0.89 vs 0.80 useful/issued) but scatters each tile's gathers over the whole feature map (L2 misses); chunks keep the
# spatial locality of the row order
MASK_SORT_CHUNK = int(__import__('os').environ.get('CRB_MASK_SORT_CHUNK', '4096'))


# output-row-window work decomposition of the wgrad (crb_sparse_conv_wgrad_windowed): L2 hit rate of the gathers 7 % -> 64 %,
# fabric reads 800 -> 350 MB per 64x64 launch, but 150 -> 196 us: the matrix pipe, not the memory side, paces the kernel once
# 3 waves share a SIMD, and 64-96 workgroup slots per XCD cannot be dealt evenly to 27 offsets. Opt-in.
WGRAD_WINDOWED = False
COMPACT_TABLES = True  # mask + packed-index tables for the kernels that have a compact instance (A/B: set False)


class CompactTable(object):
    __slots__ = ('cmask', 'cbase', 'packed', 'perm', 'n', 'K')

    def __init__(self, cmask, cbase, packed, perm, n, K):
        self.cmask, self.cbase, self.packed, self.perm, self.n, self.K = cmask, cbase, packed, perm, n, K

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


def _compact(table, perm, K):
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
    """row permutation of the gather-GEMM's kernel order (None = natural order)"""
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


def _pairs_from_nbr(nbr, n_rows, K):
    dev = nbr.device
    cap = max(n_rows * K, 1)
    pin = torch.empty((cap,), dtype=torch.int32, device=dev)
    pout = torch.empty((cap,), dtype=torch.int32, device=dev)
    pstart = torch.empty((K + 1,), dtype=torch.int32, device=dev)
    wsb = lib.crb_pairs_workspace_bytes(n_rows, K)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    check(lib.crb_pairs_from_nbr(ptr(nbr), n_rows, K, ptr(pin), ptr(pout), ptr(pstart), ptr(ws), wsb, cur_stream(dev)),
          'crb_pairs_from_nbr')
    # first pair of every output-row window per offset: the windowed wgrad's work decomposition (once per rulebook)
    bnd = torch.empty((K, lib.crb_wgrad_num_windows() + 1), dtype=torch.int32, device=dev)
    check(lib.crb_wgrad_window_bounds(ptr(pout), ptr(pstart), K, n_rows, ptr(bnd), cur_stream(dev)), 'crb_wgrad_window_bounds')
    return pin, pout, pstart, bnd


def build_hash(coords, shape):
    require_cuda(coords)
    n = coords.shape[0]
    cap = lib.crb_hash_capacity_for(n)
    hkeys = torch.empty((cap,), dtype=torch.int64, device=coords.device)
    hvals = torch.empty((cap,), dtype=torch.int32, device=coords.device)
    check(lib.crb_sparse_hash_build(ptr(coords), n, host_i32x3(shape), ptr(hkeys), ptr(hvals), cap,
                                    cur_stream(coords.device)), 'crb_sparse_hash_build')
    return hkeys, hvals, cap


def subm_rulebook(coords, shape, ksize):
    """coords (N,4) i32 cuda contiguous [b,z,y,x]"""
    require_cuda(coords)
    assert coords.dtype == torch.int32 and coords.is_contiguous()
    ksize = _triple(ksize)
    n = coords.shape[0]
    K = ksize[0] * ksize[1] * ksize[2]
    hkeys, hvals, cap = build_hash(coords, shape)
    nbr = torch.empty((n, K), dtype=torch.int32, device=coords.device)
    check(lib.crb_subm_rulebook(ptr(coords), n, host_i32x3(shape), host_i32x3(ksize), ptr(hkeys), ptr(hvals), cap,
                                ptr(nbr), cur_stream(coords.device)), 'crb_subm_rulebook')
    rb = Rulebook(nbr, None, n, n, ksize, [1, 1, 1], [k // 2 for k in ksize], True, list(shape), list(shape), coords)
    rb.in_coords = coords
    return rb


def strided_chain_counts(coords, shape, batch_size, geoms):
    """output bitmaps and output-site counts of a CHAIN of strided convs (geoms = [(ksize, stride, padding), ...], each
    applied to the previous one's output) with ONE host read-back: level 1 is marked from the coordinates, every further
    level straight from the previous level's bitmap. -> [(bitmap int32 tensor, n_out int, out_shape)] per level, to be
    handed to spconv_rulebook(..., premarked=...) which then needs no synchronisation of its own."""
    require_cuda(coords)
    dev = coords.device
    st = cur_stream(dev)
    counts = torch.empty((len(geoms),), dtype=torch.int32, device=dev)
    out, in_shape, in_bitmap = [], list(shape), None
    for lvl, (ks, sd, pd) in enumerate(geoms):
        ks, sd, pd = _triple(ks), _triple(sd), _triple(pd)
        oshape = conv_out_shape(in_shape, ks, sd, pd)
        if min(oshape) <= 0:
            raise CrbHipError(f'sparse conv output shape {oshape} is empty')
        oc = host_i32x3(oshape)
        words = lib.crb_spconv_bitmap_words(batch_size, oc)
        bitmap = torch.empty((words,), dtype=torch.int32, device=dev)
        if in_bitmap is None:
            check(lib.crb_spconv_mark(ptr(coords), coords.shape[0], batch_size, host_i32x3(ks), host_i32x3(sd),
                                      host_i32x3(pd), oc, ptr(bitmap), st), 'crb_spconv_mark')
        else:
            check(lib.crb_spconv_mark_from_bitmap(ptr(in_bitmap), batch_size, host_i32x3(in_shape), host_i32x3(ks),
                                                  host_i32x3(sd), host_i32x3(pd), oc, ptr(bitmap), st),
                  'crb_spconv_mark_from_bitmap')
        check(lib.crb_bitmap_count(ptr(bitmap), words, ptr(counts[lvl:lvl + 1]), st), 'crb_bitmap_count')
        out.append([bitmap, None, oshape])
        in_shape, in_bitmap = oshape, bitmap
    host = counts.cpu().tolist()                       # the single read-back of the chain
    for lvl, n_out in enumerate(host):
        out[lvl][1] = int(n_out)
    return [tuple(o) for o in out]


def spconv_rulebook(coords, shape, batch_size, ksize, stride, padding, premarked=None):
    """premarked = (bitmap, n_out, out_shape) from strided_chain_counts: the output set is already marked and counted, no
    host synchronisation happens here"""
    require_cuda(coords)
    assert coords.dtype == torch.int32 and coords.is_contiguous()
    ksize, stride, padding = _triple(ksize), _triple(stride), _triple(padding)
    dev = coords.device
    n = coords.shape[0]
    K = ksize[0] * ksize[1] * ksize[2]
    out_shape = conv_out_shape(shape, ksize, stride, padding)
    if min(out_shape) <= 0:
        raise CrbHipError(f'sparse conv output shape {out_shape} is empty')
    oshape_c = host_i32x3(out_shape)
    words = lib.crb_spconv_bitmap_words(batch_size, oshape_c)
    prefix = torch.empty((words,), dtype=torch.int32, device=dev)
    scan_tmp = torch.empty((words // 2048 + 2,), dtype=torch.int32, device=dev)
    n_out_dev = torch.empty((1,), dtype=torch.int32, device=dev)
    st = cur_stream(dev)
    if premarked is not None:
        bitmap, n_out, pshape = premarked
        assert list(pshape) == list(out_shape) and bitmap.numel() == words
        out_coords = torch.empty((max(n_out, 1), 4), dtype=torch.int32, device=dev)
        check(lib.crb_spconv_out_coords_premarked(batch_size, oshape_c, ptr(bitmap), ptr(prefix), ptr(scan_tmp),
                                                  ptr(out_coords), n_out, ptr(n_out_dev), st),
              'crb_spconv_out_coords_premarked')
    else:
        bitmap = torch.empty((words,), dtype=torch.int32, device=dev)
        max_out = min(max(n * K, 1), batch_size * out_shape[0] * out_shape[1] * out_shape[2])
        # upper bound used for the coordinate buffer: an input site feeds at most prod(ceil(k/s)) outputs
        fan = 1
        for k, s in zip(ksize, stride):
            fan *= (k + s - 1) // s
        max_out = min(max_out, max(n * fan, 1))
        out_coords = torch.empty((max_out, 4), dtype=torch.int32, device=dev)
        check(lib.crb_spconv_out_coords(ptr(coords), n, batch_size, host_i32x3(ksize), host_i32x3(stride),
                                        host_i32x3(padding), oshape_c, ptr(bitmap), ptr(prefix), ptr(scan_tmp),
                                        ptr(out_coords), max_out, ptr(n_out_dev), st), 'crb_spconv_out_coords')
        n_out = int(n_out_dev.item())      # sync: the output row count sizes every later buffer
        assert n_out <= max_out
    out_coords = out_coords[:n_out]
    nbr = torch.empty((n_out, K), dtype=torch.int32, device=dev)
    nbr_t = torch.empty((n, K), dtype=torch.int32, device=dev)
    check(lib.crb_spconv_rulebook(ptr(coords), n, batch_size, host_i32x3(ksize), host_i32x3(stride),
                                  host_i32x3(padding), oshape_c, ptr(bitmap), ptr(prefix), n_out, ptr(nbr), ptr(nbr_t),
                                  st), 'crb_spconv_rulebook')
    rb = Rulebook(nbr, nbr_t, n, n_out, ksize, stride, padding, False, list(shape), out_shape, out_coords)
    rb.in_coords = coords
    return rb


# When set to a list, every gather-GEMM launch appends (kind, cin, cout, K, n_in, n_out, nbr, ev0, ev1): HIP events on
# the launch stream (torch's current stream IS the stream handed to the C-ABI), read back by bench.py for the roofline.
PROFILE = None
# Arithmetic contract of the gather-GEMM (forward and dgrad) is an ARGUMENT of the call (sparse_conv(..., arithmetic=...);
# spconv.pytorch.SparseConvolution.arithmetic on the module side), never process state. 'f32' (default): exact f32 MFMA.
# 'bf16x3' (OPT-IN): operands split into two bf16 values, three bf16 MFMA passes, f32 accumulation:
# |y - y_f32| <= 2^-16 sum |x||w| (include/crb_hip.h, crb_sparse_conv_forward_bf16x3); shapes without a bf16x3 instance
# (C <= 16) keep the f32 kernel.
ARITHMETICS = ('f32', 'bf16x3')


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
            ptr(x), ptr(w_kio), ptr(table.cmask), ptr(table.cbase), ptr(table.packed), ptr(table.perm), ptr(y), n_out, K, cin,
            cout, ptr(bias.contiguous()) if bias is not None else None, ptr(bn.weight.contiguous()), ptr(bn.bias.contiguous()),
            ptr(bn.running_mean.contiguous()), ptr(bn.running_var.contiguous()), float(bn.eps), int(bool(relu)),
            cur_stream(x.device)), 'crb_sparse_conv_forward_compact_bn')
        nbr = table
        kind = kind + '_bn'
    elif arithmetic == 'bf16x3' and isinstance(table, CompactTable) and lib.crb_sparse_conv_bf16x3_supported(cin, cout):
        wsb = lib.crb_sparse_conv_bf16x3_workspace_bytes(K, cin, cout)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=x.device)
        check(lib.crb_sparse_conv_forward_bf16x3(ptr(x), ptr(w_kio), ptr(table.cmask), ptr(table.cbase), ptr(table.packed),
                                                 ptr(table.perm), ptr(y), x.shape[0], n_out, K, cin, cout, ptr(ws), wsb,
                                                 cur_stream(x.device)), 'crb_sparse_conv_forward_bf16x3')
        nbr = table
        kind = kind + '_bf16x3'
    elif isinstance(table, CompactTable):
        check(lib.crb_sparse_conv_forward_compact(ptr(x), ptr(w_kio), ptr(table.cmask), ptr(table.cbase), ptr(table.packed),
                                                  ptr(table.perm), ptr(y), n_out, K, cin, cout, cur_stream(x.device)),
              'crb_sparse_conv_forward_compact')
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
                wd = w.flip(0).transpose(1, 2).contiguous()      # Wd[o] = W[K-1-o]^T
                dx = _conv_forward_raw(dy, wd, rb.table_for('nbr', wd.shape[1], wd.shape[2], ar), rb.n_in, 'subm_dgrad',
                                       arithmetic=ar)
            else:
                wd = w.transpose(1, 2).contiguous()
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
