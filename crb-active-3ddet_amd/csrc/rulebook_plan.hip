// Fused construction of everything the sparse-conv kernels read (row a4 of SURVEY §8), sized for a whole backbone per call.
//
// Replaces the indice-pair generation inside spconv.pytorch SubMConv3d / SparseConv3d as configured by the reference at
// pcdet/models/backbones_3d/spconv_backbone.py:77-117 (third-party spconv-cu113 v2.1.21, absent from the reference tree;
// semantics restated in SURVEY Appendix A). rulebook.hip holds the one-table-at-a-time building blocks (kept for callers that
// need a single piece); this file is what the training / scoring step runs: ~26 launches and ONE host read-back for the 8
// rulebooks (12 kernel tables) of VoxelBackBone8x instead of ~230 launches.
//
//   chain   crb_spconv_chain_mark : the output sets of a CHAIN of strided convs as bitmaps — level 1 marked from the input
//           coordinates, every further level from the previous level's bitmap; the pass that reads a bitmap also leaves
//           its per-tile popcounts, one small kernel scans them for all levels -> counts (the single read-back)
//           crb_spconv_chain_emit : per level ONE kernel: coordinates in ascending (b,z,y,x) order + a rank table
//           {bits, exclusive popcount} per bitmap word (site -> row = one 8-byte load)
//   rows    crb_subm_rows / crb_spconv_rows / crb_table_masks : ONE kernel per table: neighbour rows (n,K), the per-row
//           mask of present offsets and per-4096-row-chunk counts of every offset. A stride-s conv needs no existence test
//           (every site an input reaches IS an output) and only the offsets of matching parity do a rank lookup; SubM on a
//           set that came out of the chain looks sites up in the rank table (no hash), on a foreign set in a site hash.
//   finish  crb_tables_finish : for ALL tables of the backbone in TWO launches: per 4096-row chunk one workgroup — mask sort
//           (rarest offset first), kernel-order masks, prefix of their popcounts, packed neighbour indices, tile weights and
//           the wgrad pair lists (offset-major, ascending output row) of its rows; then per (table, XCD range) the
//           heaviest-first order of the 64-row tiles, which the gather-GEMM reads as an indirection (nothing is permuted).
#include "crb_common.h"
#include "../../include/crb_hip.h"

namespace {

constexpr int CHUNK = 4096;          // rows per sort chunk (= crb_mask_sort_chunk_rows())
constexpr int TILE_WORDS = 2048;     // bitmap words per scan tile (256 threads x 8 words)
constexpr int ROWS_PER_WG = 64;      // rows kernels: 8 row slots x 8 iterations, all lookups of a lane in flight at once (divides CHUNK)

struct Shape3 { int d, h, w; };
struct ConvGeom {
  int kd, kh, kw;
  int sd, sh, sw;
  int pd, ph, pw;
};

// site key of the hash path (must equal rulebook.hip's): linear index over the true volume
__device__ __forceinline__ int64_t lin_index(int b, int z, int y, int x, Shape3 s) {
  return (((int64_t)b * s.d + z) * s.h + y) * (int64_t)s.w + x;
}

// Bitmaps of this file are ROW-PADDED: every x-row starts a new 32-bit word (ww = ceil(w / 32) words per row), so that the
// next level's row is a pure function of whole words of this level's rows (no row straddles a word) and a word decodes to
// (b, z, y, 32 x-sites) with 32-bit arithmetic. word = ((b d + z) h + y) ww + x / 32, bit = x % 32.
__device__ __forceinline__ int words_per_row(int w) { return (w + 31) >> 5; }
__device__ __forceinline__ int64_t word_index(int b, int z, int y, int x, Shape3 s) {
  return (((int64_t)b * s.d + z) * s.h + y) * words_per_row(s.w) + (x >> 5);
}

__device__ __forceinline__ int rank_lookup(const uint2* __restrict__ rank, int64_t word, int x) {
  const uint2 e = rank[word];
  const unsigned bit = (unsigned)(x & 31);
  return ((e.x >> bit) & 1u) ? (int)(e.y + __popc(e.x & ((1u << bit) - 1u))) : -1;
}

// ------------------------------------------------------------------------------------------------ chain of output sets

// level 1 (from the coordinate list, atomics): outputs reached by one input site — per axis the taps k with (c + p - k) % s == 0
// (n_dev != nullptr: the row count lives on the device - the voxel generator's total, not read back yet; n is then the capacity)
__global__ __launch_bounds__(256) void chain_mark_coords_kernel(const int* __restrict__ coords, int n, const int* __restrict__ n_dev,
                                                                ConvGeom g, Shape3 so, uint32_t* __restrict__ bitmap) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (n_dev) n = min(n, *n_dev);
  if (j >= n) return;
  const int4 c = *reinterpret_cast<const int4*>(coords + (int64_t)j * 4);
  for (int kz = 0; kz < g.kd; ++kz) {
    const int tz = c.y + g.pd - kz;
    if (tz < 0 || tz % g.sd || tz / g.sd >= so.d) continue;
    for (int ky = 0; ky < g.kh; ++ky) {
      const int ty = c.z + g.ph - ky;
      if (ty < 0 || ty % g.sh || ty / g.sh >= so.h) continue;
      for (int kx = 0; kx < g.kw; ++kx) {
        const int tx = c.w + g.pw - kx;
        if (tx < 0 || tx % g.sw || tx / g.sw >= so.w) continue;
        const int ox = tx / g.sw;
        atomicOr(&bitmap[word_index(c.x, tz / g.sd, ty / g.sh, ox, so)], 1u << (ox & 31));
      }
    }
  }
}

__device__ __forceinline__ uint32_t even_bits(unsigned long long x) {         // bit j of the result = bit 2j of x
  x &= 0x5555555555555555ULL;
  x = (x | (x >> 1)) & 0x3333333333333333ULL;
  x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0fULL;
  x = (x | (x >> 4)) & 0x00ff00ff00ff00ffULL;
  x = (x | (x >> 8)) & 0x0000ffff0000ffffULL;
  x = (x | (x >> 16)) & 0x00000000ffffffffULL;
  return (uint32_t)x;
}

// 32 output x-sites (word wx of an output row) from one input row: out[ox] = OR_kx in[ox sw - pw + kx]
__device__ __forceinline__ uint32_t row_downsample(const uint32_t* __restrict__ row, int wwi, int wi, int wx, int kw, int sw,
                                                   int pw) {
  if (kw == 3 && sw == 2 && pw == 1) {            // in bits 64 wx + 2 j + {-1, 0, 1}
    const int w0 = 2 * wx;
    const unsigned long long lo = w0 < wwi ? row[w0] : 0u, hi = w0 + 1 < wwi ? row[w0 + 1] : 0u;
    const unsigned long long in = (hi << 32) | lo;
    const unsigned long long prev = (w0 > 0 && w0 - 1 < wwi) ? (row[w0 - 1] >> 31) : 0u;
    return even_bits(in | (in >> 1) | ((in << 1) | prev));
  }
  if (kw == 1 && sw == 1 && pw == 0) return wx < wwi ? row[wx] : 0u;
  uint32_t out = 0;                               // any other geometry: bit by bit
  for (int j = 0; j < 32; ++j)
    for (int kx = 0; kx < kw; ++kx) {
      const int ix = (wx * 32 + j) * sw - pw + kx;
      if (ix >= 0 && ix < wi) out |= ((row[ix >> 5] >> (ix & 31)) & 1u) << j;
    }
  return out;
}

// levels >= 2, output-stationary, no atomics and no zero-fill: one 1024-thread workgroup per 2048-word tile of the OUTPUT
// bitmap computes every word from the <= kd*kh input rows above it (two words per thread, all row loads of a thread
// independent: the pass is bound by load latency) and leaves the tile's popcount. in_bitmap == nullptr: count only (the level
// marked from the coordinate list).
__global__ __launch_bounds__(1024) void chain_mark_bitmap_kernel(const uint32_t* __restrict__ in_bitmap, Shape3 si, ConvGeom g,
                                                                 Shape3 so, int B, uint32_t* __restrict__ out_bitmap,
                                                                 int* __restrict__ tile_sums) {
  __shared__ int sh[16];
  const int wwo = words_per_row(so.w), wwi = words_per_row(si.w);
  const int64_t words = (int64_t)B * so.d * so.h * wwo;
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int64_t wo = (int64_t)blockIdx.x * TILE_WORDS + k * 1024 + threadIdx.x;
    uint32_t m = 0;
    if (in_bitmap == nullptr) {
      m = out_bitmap[wo];
    } else {
      if (wo < words) {
        const unsigned row = (unsigned)(wo / wwo);
        const int wx = (int)(wo - (int64_t)row * wwo);
        const int oy = (int)(row % (unsigned)so.h);
        const unsigned t = row / (unsigned)so.h;
        const int oz = (int)(t % (unsigned)so.d), b = (int)(t / (unsigned)so.d);
        if (g.kd <= 3 && g.kh <= 3) {
          uint32_t part[9];
#pragma unroll
          for (int kz = 0; kz < 3; ++kz)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
              const int iz = oz * g.sd - g.pd + kz, iy = oy * g.sh - g.ph + ky;
              part[kz * 3 + ky] = 0;
              if (kz < g.kd && ky < g.kh && iz >= 0 && iz < si.d && iy >= 0 && iy < si.h)
                part[kz * 3 + ky] = row_downsample(in_bitmap + (((int64_t)b * si.d + iz) * si.h + iy) * wwi, wwi, si.w, wx,
                                                   g.kw, g.sw, g.pw);
            }
#pragma unroll
          for (int q = 0; q < 9; ++q) m |= part[q];
        } else {
          for (int kz = 0; kz < g.kd; ++kz) {
            const int iz = oz * g.sd - g.pd + kz;
            if (iz < 0 || iz >= si.d) continue;
            for (int ky = 0; ky < g.kh; ++ky) {
              const int iy = oy * g.sh - g.ph + ky;
              if (iy < 0 || iy >= si.h) continue;
              m |= row_downsample(in_bitmap + (((int64_t)b * si.d + iz) * si.h + iy) * wwi, wwi, si.w, wx, g.kw, g.sw, g.pw);
            }
          }
        }
        const int valid = so.w - wx * 32;                        // x-sites of this word inside the volume
        if (valid < 32) m &= (1u << valid) - 1u;
      }
      out_bitmap[wo] = m;                                        // also zero-fills the padding words of the tile
    }
    cnt += __popc(m);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += sh[w];
    tile_sums[blockIdx.x] = tot;
  }
}

struct ChainScanArgs {
  int tile_off[9];      // first tile of level l in tile_sums (l = 0..L)
  int levels;
};

// one workgroup per level: exclusive scan of the level's tile popcounts in place, total -> counts[level]
__global__ __launch_bounds__(256) void chain_scan_kernel(int* __restrict__ tile_sums, ChainScanArgs a, int* __restrict__ counts) {
  __shared__ int sh[4];
  const int l = blockIdx.x;
  int* ts = tile_sums + a.tile_off[l];
  const int m = a.tile_off[l + 1] - a.tile_off[l];
  int carry = 0;
  for (int base = 0; base < m; base += 256) {
    const int i = base + (int)threadIdx.x;
    const int v = (i < m) ? ts[i] : 0;
    int tot;
    const int ex = crb_block_excl_scan_256(v, sh, &tot);
    if (i < m) ts[i] = carry + ex;
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[l] = carry;
}

// one workgroup per tile: rank table {bits, exclusive popcount} per word + the coordinates of the set bits, rows in
// ascending (b,z,y,x) order (= ascending word / bit order of the row-padded bitmap)
__global__ __launch_bounds__(256) void chain_emit_kernel(const uint32_t* __restrict__ bitmap, const int* __restrict__ tile_prefix,
                                                         Shape3 so, int max_out, uint2* __restrict__ rank,
                                                         int* __restrict__ out_coords) {
  __shared__ int sh[4];
  const int64_t w0 = (int64_t)blockIdx.x * TILE_WORDS + (int64_t)threadIdx.x * 8;
  const uint4 a = *reinterpret_cast<const uint4*>(bitmap + w0);
  const uint4 b4 = *reinterpret_cast<const uint4*>(bitmap + w0 + 4);
  const uint32_t wds[8] = {a.x, a.y, a.z, a.w, b4.x, b4.y, b4.z, b4.w};
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) cnt += __popc(wds[k]);
  int tot;
  int r = crb_block_excl_scan_256(cnt, sh, &tot) + tile_prefix[blockIdx.x];
  const int wwo = words_per_row(so.w);
  uint2 e[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    e[k] = make_uint2(wds[k], (unsigned)r);
    uint32_t m = wds[k];
    if (m) {
      const unsigned row = (unsigned)((w0 + k) / wwo);
      const int x0 = (int)((w0 + k) - (int64_t)row * wwo) * 32;
      const int y = (int)(row % (unsigned)so.h);
      const unsigned t = row / (unsigned)so.h;
      const int z = (int)(t % (unsigned)so.d), b = (int)(t / (unsigned)so.d);
      while (m) {
        const int bit = __ffs(m) - 1;
        m &= m - 1;
        if (r < max_out) *reinterpret_cast<int4*>(out_coords + (int64_t)r * 4) = make_int4(b, z, y, x0 + bit);
        ++r;
      }
    }
  }
  uint4* dst = reinterpret_cast<uint4*>(rank + w0);
#pragma unroll
  for (int k = 0; k < 4; ++k) dst[k] = make_uint4(e[2 * k].x, e[2 * k].y, e[2 * k + 1].x, e[2 * k + 1].y);
}

// ------------------------------------------------------------------------------------------------ neighbour rows

struct RowsArgs {
  const int* coords;        // (n,4) the rows of this table
  int n, K, mode;           // mode 0 SubM via site hash, 1 SubM via rank table, 2 strided (input-stationary), 3 table -> masks
  Shape3 s;                 // SubM: the volume; strided: the OUTPUT volume
  ConvGeom g;
  const long long* hkeys;
  const int* hvals;
  uint32_t hmask;
  const uint2* rank;        // mode 1: of this set; mode 2: of the output set
  int n_lookup;             // rows of the looked-up set
  int* nbr;                 // (n,K): SubM table / strided nbr_t / mode 3: the table read
  int* scatter;             // mode 2: nbr (n_out,K), pre-filled with -1
  unsigned* mask;           // (n) or null
  int* hist;                // (chunks,32), zero-filled by the caller, or null
};

// t = s * q exactly? -> q (else -1); strides 1 and 2 without a division
__device__ __forceinline__ int exact_div(int t, int s) {
  if (s == 1) return t;
  if (s == 2) return (t & 1) ? -1 : (t >> 1);
  return (t % s) ? -1 : t / s;
}

// where a lane has to look: `word` of the rank table + x (modes 1, 2) or the site key (mode 0); -1 = outside / wrong parity
__device__ __forceinline__ int64_t row_target(const RowsArgs& a, int kz, int ky, int kx, int4 c, int* xbit) {
  const ConvGeom g = a.g;
  if (a.mode == 2) {
    const int tz = c.y + g.pd - kz, ty = c.z + g.ph - ky, tx = c.w + g.pw - kx;
    if (tz < 0 || ty < 0 || tx < 0) return -1;
    const int oz = exact_div(tz, g.sd), oy = exact_div(ty, g.sh), ox = exact_div(tx, g.sw);
    if (oz < 0 || oy < 0 || ox < 0 || oz >= a.s.d || oy >= a.s.h || ox >= a.s.w) return -1;
    *xbit = ox;
    return word_index(c.x, oz, oy, ox, a.s);
  }
  const int z = c.y + kz - g.kd / 2, y = c.z + ky - g.kh / 2, x = c.w + kx - g.kw / 2;
  if (z < 0 || z >= a.s.d || y < 0 || y >= a.s.h || x < 0 || x >= a.s.w) return -1;
  *xbit = x;
  return a.mode == 1 ? word_index(c.x, z, y, x, a.s) : lin_index(c.x, z, y, x, a.s);
}

// 32 lanes per row (lane = kernel offset), 8 rows per workgroup iteration, 64 consecutive rows per workgroup: the lookups of
// a lane's 8 rows are independent and issued back to back (the kernel is bound by lookup latency, not by bytes); the hash
// probes of the 8 rows advance in lockstep for the same reason (8 sequential probe loops took 115 us on the level-1 set).
__global__ __launch_bounds__(256) void table_rows_kernel(RowsArgs a) {
  __shared__ int cnt_sh[8][32];
  constexpr int R = ROWS_PER_WG / 8;
  const int o = threadIdx.x & 31, slot = threadIdx.x >> 5;
  const int row0 = blockIdx.x * ROWS_PER_WG;
  const bool act = o < a.K;
  const int kx = o % a.g.kw, ky = (o / a.g.kw) % a.g.kh, kz = o / (a.g.kw * a.g.kh);
  int r[R];
  if (a.mode == 3) {
#pragma unroll
    for (int it = 0; it < R; ++it) {
      const int i = row0 + it * 8 + slot;
      r[it] = (i < a.n && act) ? a.nbr[(int64_t)i * a.K + o] : -1;
    }
  } else {
    int64_t tgt[R];
    int xb[R];
#pragma unroll
    for (int it = 0; it < R; ++it) {
      const int i = row0 + it * 8 + slot;
      tgt[it] = -1;
      xb[it] = 0;
      if (i < a.n && act) tgt[it] = row_target(a, kz, ky, kx, *reinterpret_cast<const int4*>(a.coords + (int64_t)i * 4), &xb[it]);
    }
    if (a.mode == 0) {
      const uint32_t gmask = a.hmask >> 3;
      uint32_t g[R], g0[R], sub[R];                   // sub: residue class being probed, bits 3.. = finished rounds (crb_ghash_find)
      bool open[R];
      bool any = false;
#pragma unroll
      for (int it = 0; it < R; ++it) {
        open[it] = tgt[it] >= 0;
        g0[it] = g[it] = crb_hash64(tgt[it] >> 3) & gmask;
        sub[it] = (uint32_t)(tgt[it] & 7);
        r[it] = -1;
        any |= open[it];
      }
      while (any) {                                   // one probe of every open lookup per trip, loads back to back
        long long kq[R];
#pragma unroll
        for (int it = 0; it < R; ++it) kq[it] = open[it] ? a.hkeys[(g[it] << 3) | (sub[it] & 7u)] : 0;
        any = false;
#pragma unroll
        for (int it = 0; it < R; ++it) {
          if (!open[it]) continue;
          if (kq[it] == (long long)tgt[it]) { r[it] = (int)((g[it] << 3) | (sub[it] & 7u)); open[it] = false; }
          else if (kq[it] == CRB_HASH_EMPTY) open[it] = false;
          else {
            g[it] = (g[it] + 1) & gmask;
            if (g[it] == g0[it]) {                    // a full residue class: the insert spilled into the next one
              sub[it] = ((sub[it] + 1) & 7u) | ((sub[it] & ~7u) + 8u);
              if ((sub[it] >> 3) >= 8u) { open[it] = false; continue; }
            }
            any = true;
          }
        }
      }
#pragma unroll
      for (int it = 0; it < R; ++it) if (r[it] >= 0) r[it] = a.hvals[r[it]];
    } else {
      uint2 e[R];
#pragma unroll
      for (int it = 0; it < R; ++it) e[it] = tgt[it] >= 0 ? a.rank[tgt[it]] : make_uint2(0u, 0u);
#pragma unroll
      for (int it = 0; it < R; ++it) {
        const unsigned bit = (unsigned)(xb[it] & 31);
        const int rr = (int)(e[it].y + __popc(e[it].x & ((1u << bit) - 1u)));
        r[it] = (tgt[it] >= 0 && ((e[it].x >> bit) & 1u) && rr < a.n_lookup) ? rr : -1;
      }
    }
  }
  int cnt = 0;
#pragma unroll
  for (int it = 0; it < R; ++it) {
    const int i = row0 + it * 8 + slot;
    const bool live = i < a.n;                        // uniform over the 32 lanes of a row
    if (live && act && a.mode != 3) {
      a.nbr[(int64_t)i * a.K + o] = r[it];
      if (a.mode == 2 && r[it] >= 0) a.scatter[(int64_t)r[it] * a.K + o] = i;   // unique writer: (output site, offset) -> input site
    }
    const unsigned long long b = __ballot(r[it] >= 0);
    if (live && o == 0 && a.mask) a.mask[i] = (unsigned)((threadIdx.x & 32) ? (b >> 32) : (b & 0xffffffffULL));
    cnt += r[it] >= 0;
  }
  if (a.hist == nullptr) return;
  cnt_sh[slot][o] = cnt;
  __syncthreads();
  if (threadIdx.x < 32) {
    int s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += cnt_sh[k][threadIdx.x];
    if (s) atomicAdd(&a.hist[(row0 / CHUNK) * 32 + threadIdx.x], s);
  }
}

// ------------------------------------------------------------------------------------------------ finish: sort pass + fill pass

constexpr int MAX_TABLES = 16;
CRB_KNOB g_chunk_skip = 0;      // measurement builds (wrong results): bit 0 = no sort, 1 = no packed fill, 2 = no pair lists

struct FinishArgs {
  CrbTablePlan t[MAX_TABLES];
  int chunk_start[MAX_TABLES + 1];     // sort pass: first workgroup (4096-row chunk) of table k
  int block_start[MAX_TABLES + 1];     // fill pass: first workgroup (256-row block) of table k
  int n_tables;
  int skip;                            // g_chunk_skip (0 in the product library)
};

// result of a bitonic compare-exchange for the element that holds v and sees its partner's p
__device__ __forceinline__ unsigned long long bitonic_keep(unsigned long long v, unsigned long long p, bool keep_min) {
  return ((v < p) == keep_min) ? v : p;
}

// Sort pass: one 1024-thread workgroup per chunk of 4096 consecutive rows of one table. Thread t owns the elements 4t..4t+3 of
// the bitonic network in registers: partners at distance 1, 2 are its own, at 4..128 another lane of its wave (shuffles), only
// the 10 steps at distance >= 256 go through LDS (78 barrier steps with every key in LDS took 100 us per backbone).
__global__ __launch_bounds__(1024) void tables_sort_kernel(FinishArgs fa) {
  __shared__ __attribute__((aligned(16))) unsigned long long key[CHUNK];      // 32 KB: exchange buffer of the LDS steps
  __shared__ unsigned nat_sh[CHUNK];                                          // 16 KB: masks in natural order
  __shared__ int ucnt[64][32];                       // first use: [0..31] = "before" partials, [32..63] = "total" partials
  __shared__ int lpos[32], pbase[33], wave_tot[16];
  __shared__ int chunk_base_sh;
  int k = 0;
  while (k + 1 < fa.n_tables && (int)blockIdx.x >= fa.chunk_start[k + 1]) ++k;
  const CrbTablePlan T = fa.t[k];
  const int chunk = blockIdx.x - fa.chunk_start[k];
  const int n = (int)T.n, K = T.K;
  const int nchunks = (n + CHUNK - 1) / CHUNK;
  const int base = chunk * CHUNK;
  const int rows = min(CHUNK, n - base);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- per-offset counts: of this chunk (sort key), of the chunks before it (pair positions), of the table (pair_start)
  {
    const int o = tid & 31, part = tid >> 5;
    int before = 0, total = 0;
    for (int c = part; c < nchunks; c += 32) {
      const int v = T.hist[c * 32 + o];
      total += v;
      if (c < chunk) before += v;
    }
    ucnt[part][o] = before;
    ucnt[32 + part][o] = total;
  }
  for (int t = tid; t < CHUNK; t += 1024) nat_sh[t] = t < rows ? T.mask[base + t] : 0u;
  __syncthreads();
  if (tid < 32) {
    int before = 0, total = 0;
#pragma unroll
    for (int p = 0; p < 32; ++p) { before += ucnt[p][tid]; total += ucnt[32 + p][tid]; }
    const int mine = T.hist[chunk * 32 + tid];
    int incl = total;                                 // exclusive prefix over offsets of the table totals = pair_start
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
      const int v = __shfl_up(incl, d, 64);
      if (tid >= d) incl += v;
    }
    pbase[tid] = incl - total + before;
    if (tid == 31) pbase[32] = incl;                                  // P
    if (chunk == 0 && T.pair_start) {
      if (tid <= K) T.pair_start[tid] = incl - total;                 // offsets >= K have no pairs: entry K = P
      if (K == 32 && tid == 31) T.pair_start[32] = incl;
    }
    int cb = before;                                                  // packed base of the chunk = pairs of earlier chunks
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) cb += __shfl_xor(cb, d, 64);
    if (tid == 0) chunk_base_sh = cb;
    // rank of the bits: rarest offset of THIS chunk = most significant (ties: lower bit index first)
    int p = 0;
    for (int b = 0; b < 32; ++b) {
      const int hb = __shfl(mine, b, 64);
      p += (hb > mine) || (hb == mine && b < tid);
    }
    lpos[tid] = p;
  }
  __syncthreads();

  // ---- sort keys: (~ranked mask) << 32 | local row : descending ranked mask, stable; rows beyond the table sort last
  unsigned long long v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int t = tid * 4 + q;
    v[q] = ~0ULL;
    if (t < rows) {
      const unsigned m = nat_sh[t];
      unsigned r = 0;
      for (int b = 0; b < 32; ++b) r |= ((m >> b) & 1u) << lpos[b];
      v[q] = ((unsigned long long)(~r) << 32) | (unsigned)t;
    }
  }
  if (!(fa.skip & 1)) {
    for (int kk = 2; kk <= CHUNK; kk <<= 1) {
      for (int j = kk >> 1; j >= 256; j >>= 1) {                      // partner in another wave: through LDS
        *reinterpret_cast<ulonglong2*>(&key[tid * 4]) = make_ulonglong2(v[0], v[1]);
        *reinterpret_cast<ulonglong2*>(&key[tid * 4 + 2]) = make_ulonglong2(v[2], v[3]);
        __syncthreads();
        const int pt = (tid ^ (j >> 2)) * 4;
        const ulonglong2 p01 = *reinterpret_cast<const ulonglong2*>(&key[pt]);
        const ulonglong2 p23 = *reinterpret_cast<const ulonglong2*>(&key[pt + 2]);
        const bool keep_min = (((tid * 4) & j) == 0) == (((tid * 4) & kk) == 0);
        v[0] = bitonic_keep(v[0], p01.x, keep_min);
        v[1] = bitonic_keep(v[1], p01.y, keep_min);
        v[2] = bitonic_keep(v[2], p23.x, keep_min);
        v[3] = bitonic_keep(v[3], p23.y, keep_min);
        __syncthreads();
      }
      for (int j = min(kk >> 1, 128); j >= 4; j >>= 1) {              // partner in another lane of this wave
        const bool keep_min = (((tid * 4) & j) == 0) == (((tid * 4) & kk) == 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned long long p = __shfl_xor(v[q], j >> 2, 64);
          v[q] = bitonic_keep(v[q], p, keep_min);
        }
      }
      if (kk >= 4) {                                                  // distance 2: (0,2) (1,3)
        const bool up = ((tid * 4) & kk) == 0;
        const unsigned long long a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
        v[0] = bitonic_keep(a0, a2, up);  v[2] = bitonic_keep(a2, a0, !up);
        v[1] = bitonic_keep(a1, a3, up);  v[3] = bitonic_keep(a3, a1, !up);
      }
      {                                                               // distance 1: (0,1) (2,3)
        const bool up01 = kk == 2 ? true : ((tid * 4) & kk) == 0;
        const bool up23 = kk == 2 ? false : up01;                     // kk == 2: elements 2, 3 have bit 1 set -> descending pair
        const unsigned long long a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
        v[0] = bitonic_keep(a0, a1, up01);  v[1] = bitonic_keep(a1, a0, !up01);
        v[2] = bitonic_keep(a2, a3, up23);  v[3] = bitonic_keep(a3, a2, !up23);
      }
    }
  }

  // ---- kernel order: perm, cmask, cbase (4 consecutive rows per thread)
  unsigned m4[4];
  int id4[4], pc = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int t = tid * 4 + q;
    id4[q] = (int)(v[q] & 0xffffffffULL);
    m4[q] = t < rows ? nat_sh[id4[q]] : 0u;
    pc += __popc(m4[q]);
  }
  int incl = crb_wave_incl_scan(pc);
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < wave; ++w) woff += wave_tot[w];
  int ex = chunk_base_sh + woff + incl - pc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int t = tid * 4 + q;
    if (t < rows) {
      T.perm[base + t] = base + id4[q];
      T.cmask[base + t] = m4[q];
      T.cbase[base + t] = ex;
    }
    ex += __popc(m4[q]);
  }
  if (chunk == nchunks - 1 && tid == 1023) T.cbase[n] = ex;           // = P (rows beyond n have empty masks)

  // ---- tile weights: offsets present in any of the 64 rows of a tile (16 consecutive threads)
  {
    unsigned m = m4[0] | m4[1] | m4[2] | m4[3];
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) m |= (unsigned)__shfl_xor((int)m, d, 64);
    if ((tid & 15) == 0 && tid * 4 < rows) T.tile_weight[chunk * (CHUNK / 64) + (tid >> 4)] = __popc(m);
  }

  // ---- pair positions: first pair of every (64-row unit of the natural order, offset) of this chunk
  if (T.pair_unit_base == nullptr) return;
  __syncthreads();                                                    // ucnt is reused
  for (int u = wave; u < CHUNK / 64; u += 16) {
    const unsigned m = nat_sh[u * 64 + lane];
    for (int o = 0; o < 32; ++o) {
      const int c = __popcll(__ballot((m >> o) & 1u));
      if (lane == o) ucnt[u][o] = c;
    }
  }
  __syncthreads();
  if (tid < 32) {                                                     // exclusive prefix over the 64 units, per offset
    int run = pbase[tid];
    for (int u = 0; u < CHUNK / 64; ++u) {
      const int c = ucnt[u][tid];
      ucnt[u][tid] = run;
      run += c;
    }
  }
  __syncthreads();
  const int units = (rows + 63) / 64;
  for (int t = tid; t < units * 32; t += 1024) T.pair_unit_base[(int64_t)chunk * (CHUNK / 64) * 32 + t] = ucnt[t >> 5][t & 31];
}

// Fill pass: one 256-thread workgroup per 256 rows. (1) packed neighbour indices of the kernel-order rows [256 w, 256 w + 256):
// 32 lanes per row, 8 independent gathers in flight per lane. (2) pair lists of the natural-order rows of the same range:
// one wave per 64-row unit, offset-major, ascending output row.
__global__ __launch_bounds__(256) void tables_fill_kernel(FinishArgs fa) {
  __shared__ int ub_sh[4][32];
  int k = 0;
  while (k + 1 < fa.n_tables && (int)blockIdx.x >= fa.block_start[k + 1]) ++k;
  const CrbTablePlan T = fa.t[k];
  const int blk = blockIdx.x - fa.block_start[k];
  const int n = (int)T.n, K = T.K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (!(fa.skip & 2)) {
    const int o = tid & 31, oc = o < K ? o : K - 1, slot = tid >> 5;
    for (int it0 = 0; it0 < 32; it0 += 8) {
      unsigned mm[8];
      int cb[8], v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int t = blk * 256 + (it0 + q) * 8 + slot;
        mm[q] = 0u;
        if (t < n) {
          mm[q] = T.cmask[t];
          cb[q] = T.cbase[t];
          v[q] = T.nbr[(int64_t)T.perm[t] * K + oc];
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if ((mm[q] >> o) & 1u) T.packed[cb[q] + __popc(mm[q] & ((1u << o) - 1u))] = v[q];
    }
  }
  if (T.pair_in == nullptr || (fa.skip & 4)) return;
  const int u = blk * 4 + wave;
  const int r = u * 64 + lane;
  if (u * 64 >= n) return;
  if (lane < 32) ub_sh[wave][lane] = T.pair_unit_base[(int64_t)u * 32 + lane];
  const unsigned m = r < n ? T.mask[r] : 0u;
  const int* row = T.nbr + (int64_t)(r < n ? r : 0) * K;
  int v[32];
#pragma unroll
  for (int o = 0; o < 32; ++o) v[o] = row[o < K ? o : K - 1];          // the row's entries, all loads in flight at once
#pragma unroll
  for (int o = 0; o < 32; ++o) {
    const unsigned long long b = __ballot((m >> o) & 1u);
    if ((m >> o) & 1u) {
      const int pos = ub_sh[wave][o] + __popcll(b & ((1ULL << lane) - 1ULL));
      T.pair_in[pos] = v[o];
      T.pair_out[pos] = r;
    }
  }
}

// Heaviest-tile-first order of the 64-row tiles inside each of the 8 contiguous tile ranges (one per XCD: workgroup b of the
// gather-GEMM runs on XCD b % 8 and takes position (b % 8) * per + b / 8). Stable (ties keep table order): deterministic.
// One workgroup per (table, range); ranges of up to 8192 tiles are sorted (keys in LDS), longer ones keep table order.
constexpr int LPT_RANGES = 8;
constexpr int LPT_MAX = 8192;

__global__ __launch_bounds__(1024) void tables_order_kernel(FinishArgs fa) {
  __shared__ unsigned key[LPT_MAX];
  const int k = blockIdx.x / LPT_RANGES, r = blockIdx.x % LPT_RANGES;
  const CrbTablePlan T = fa.t[k];
  const int n = (int)T.n;
  const int tiles_all = (n + 63) / 64, tiles_full = n / 64;
  const int per = (tiles_all + LPT_RANGES - 1) / LPT_RANGES;
  const int lo = r * per, hi = min(lo + per, tiles_all);
  if (lo >= hi) return;
  const int hi_full = min(hi, tiles_full);                           // the trailing partial tile keeps its (last) place
  const int cnt = max(hi_full - lo, 0);
  if (hi > hi_full && threadIdx.x == 0) T.tile_order[hi_full] = hi_full;
  if (cnt == 0) return;
  if (cnt > LPT_MAX) {
    for (int t = threadIdx.x; t < cnt; t += 1024) T.tile_order[lo + t] = lo + t;
    return;
  }
  int p2 = 2;
  while (p2 < cnt) p2 <<= 1;
  for (int t = threadIdx.x; t < p2; t += 1024)
    key[t] = t < cnt ? ((unsigned)(32 - T.tile_weight[lo + t]) << 16) | (unsigned)t : 0xffffffffu;
  __syncthreads();
  for (int kk = 2; kk <= p2; kk <<= 1)
    for (int j = kk >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < p2 / 2; t += 1024) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int p = i | j;
        const bool up = (i & kk) == 0;
        const unsigned a = key[i], b = key[p];
        if ((a > b) == up) { key[i] = b; key[p] = a; }
      }
      __syncthreads();
    }
  for (int t = threadIdx.x; t < cnt; t += 1024) T.tile_order[lo + t] = lo + (int)(key[t] & 0xffffu);
}

}  // namespace

// ================================================================================================ C-ABI

extern "C" int64_t crb_spconv_padded_words(int B, const int32_t* out_shape_dhw) {
  const int64_t words = (int64_t)B * out_shape_dhw[0] * out_shape_dhw[1] * ((out_shape_dhw[2] + 31) / 32);   // row-padded
  return crb_align_up(words, TILE_WORDS);
}

static inline ConvGeom geom_of(const int32_t* g9) { return ConvGeom{g9[0], g9[1], g9[2], g9[3], g9[4], g9[5], g9[6], g9[7], g9[8]}; }
static inline Shape3 shape_of(const int32_t* s) { return Shape3{s[0], s[1], s[2]}; }

static int chain_mark_impl(const int32_t* coords, int64_t n, const int32_t* n_dev, int B, const int32_t* in_shape_dhw, int n_levels,
                           const int32_t* geoms, const int32_t* out_shapes, const int64_t* word_off, uint32_t* bitmap_all,
                           int32_t* tile_sums_all, int32_t* counts_dev, void* stream) {
  if (n < 0 || B <= 0 || n_levels < 1 || n_levels > 8) return CRB_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  for (int l = 0; l <= n_levels; ++l)
    if (word_off[l] % TILE_WORDS || (l && word_off[l] - word_off[l - 1] < crb_spconv_padded_words(B, out_shapes + 3 * (l - 1))))
      return CRB_ERR_ARG;
  if (word_off[n_levels] >= (1LL << 31)) return CRB_ERR_UNSUPPORTED;          // 32-bit word decode
  ChainScanArgs sa;
  sa.levels = n_levels;
  for (int l = 0; l <= n_levels; ++l) sa.tile_off[l] = (int)((word_off[l] - word_off[0]) / TILE_WORDS);
  // level 1: zero-fill + input-stationary marking from the coordinate list + tile popcounts
  CRB_HIP(hipMemsetAsync(bitmap_all + word_off[0], 0, (size_t)(word_off[1] - word_off[0]) * 4, st));
  if (n > 0)
    hipLaunchKernelGGL(chain_mark_coords_kernel, dim3(crb_cdiv(n, 256)), dim3(256), 0, st, coords, (int)n, n_dev, geom_of(geoms),
                       shape_of(out_shapes), bitmap_all + word_off[0]);
  hipLaunchKernelGGL(chain_mark_bitmap_kernel, dim3(sa.tile_off[1] - sa.tile_off[0]), dim3(1024), 0, st,
                     (const uint32_t*)nullptr, shape_of(out_shapes), geom_of(geoms), shape_of(out_shapes), B,
                     bitmap_all + word_off[0], tile_sums_all + sa.tile_off[0]);
  // levels 2..: output-stationary from the previous level's bitmap (writes every word of its tiles: no zero-fill)
  for (int l = 1; l < n_levels; ++l)
    hipLaunchKernelGGL(chain_mark_bitmap_kernel, dim3(sa.tile_off[l + 1] - sa.tile_off[l]), dim3(1024), 0, st,
                       (const uint32_t*)(bitmap_all + word_off[l - 1]), shape_of(out_shapes + 3 * (l - 1)), geom_of(geoms + 9 * l),
                       shape_of(out_shapes + 3 * l), B, bitmap_all + word_off[l], tile_sums_all + sa.tile_off[l]);
  hipLaunchKernelGGL(chain_scan_kernel, dim3(n_levels), dim3(256), 0, st, tile_sums_all, sa, counts_dev);
  CRB_CHECK_LAUNCH();
  (void)in_shape_dhw;
  return CRB_OK;
}

extern "C" int crb_spconv_chain_mark(const int32_t* coords, int64_t n, int B, const int32_t* in_shape_dhw, int n_levels,
                                     const int32_t* geoms, const int32_t* out_shapes, const int64_t* word_off,
                                     uint32_t* bitmap_all, int32_t* tile_sums_all, int32_t* counts_dev, void* stream) {
  return chain_mark_impl(coords, n, nullptr, B, in_shape_dhw, n_levels, geoms, out_shapes, word_off, bitmap_all, tile_sums_all,
                         counts_dev, stream);
}

// the same with the input row count still on the device: coords holds n_cap rows of which the first *n_dev are valid (the voxel
// generator's coordinate buffer and its total, crb_voxelize's counts[B]) - the chain can be marked and counted BEFORE the host
// knows the voxel count, and ONE read-back then returns it together with the level sizes
extern "C" int crb_spconv_chain_mark_lazy(const int32_t* coords, int64_t n_cap, const int32_t* n_dev, int B, const int32_t* in_shape_dhw,
                                          int n_levels, const int32_t* geoms, const int32_t* out_shapes, const int64_t* word_off,
                                          uint32_t* bitmap_all, int32_t* tile_sums_all, int32_t* counts_dev, void* stream) {
  if (!n_dev) return CRB_ERR_ARG;
  return chain_mark_impl(coords, n_cap, n_dev, B, in_shape_dhw, n_levels, geoms, out_shapes, word_off, bitmap_all, tile_sums_all,
                         counts_dev, stream);
}

extern "C" int crb_spconv_chain_emit(int B, int n_levels, const int32_t* out_shapes, const int64_t* word_off,
                                     const uint32_t* bitmap_all, const int32_t* tile_sums_all, void* rank_all,
                                     int32_t* const* out_coords, const int64_t* n_out, void* stream) {
  if (B <= 0 || n_levels < 1 || n_levels > 8) return CRB_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  for (int l = 0; l < n_levels; ++l) {
    const int tiles = (int)((word_off[l + 1] - word_off[l]) / TILE_WORDS);
    const int64_t cap = n_out[l] > 0x7fffffff ? 0x7fffffff : n_out[l];
    hipLaunchKernelGGL(chain_emit_kernel, dim3(tiles), dim3(256), 0, st, bitmap_all + word_off[l],
                       tile_sums_all + (word_off[l] - word_off[0]) / TILE_WORDS, shape_of(out_shapes + 3 * l), (int)cap,
                       (uint2*)rank_all + word_off[l], out_coords[l]);
  }
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_table_chunk_rows(void) { return CHUNK; }

static int launch_rows(const RowsArgs& a, hipStream_t st) {
  if (a.n <= 0) return CRB_OK;
  if (a.K <= 0 || a.K > 32) return CRB_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(table_rows_kernel, dim3(crb_cdiv(a.n, ROWS_PER_WG)), dim3(256), 0, st, a);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_subm_rows(const int32_t* coords, int64_t n, const int32_t* shape_dhw, const int32_t* ksize,
                             const int64_t* hkeys, const int32_t* hvals, int64_t capacity, const void* rank,
                             int32_t* nbr, uint32_t* mask, int32_t* hist, void* stream) {
  if (n < 0 || n >= (1LL << 31)) return CRB_ERR_ARG;
  if (n == 0) return CRB_OK;
  if (!(ksize[0] & 1) || !(ksize[1] & 1) || !(ksize[2] & 1)) return CRB_ERR_UNSUPPORTED;
  if (!rank && (!hkeys || !hvals || capacity <= 0 || (capacity & (capacity - 1)))) return CRB_ERR_ARG;
  RowsArgs a{};
  a.coords = coords; a.n = (int)n; a.K = ksize[0] * ksize[1] * ksize[2]; a.mode = rank ? 1 : 0;
  a.s = shape_of(shape_dhw);
  a.g = ConvGeom{ksize[0], ksize[1], ksize[2], 1, 1, 1, ksize[0] / 2, ksize[1] / 2, ksize[2] / 2};
  a.hkeys = (const long long*)hkeys; a.hvals = hvals; a.hmask = (uint32_t)(capacity - 1);
  a.rank = (const uint2*)rank; a.n_lookup = (int)n;
  a.nbr = nbr; a.scatter = nullptr; a.mask = mask; a.hist = hist;
  return launch_rows(a, (hipStream_t)stream);
}

extern "C" int crb_spconv_rows(const int32_t* coords, int64_t n, const int32_t* ksize, const int32_t* stride,
                               const int32_t* padding, const int32_t* out_shape_dhw, const void* rank_out, int64_t n_out,
                               int32_t* nbr, int32_t* nbr_t, uint32_t* mask_t, int32_t* hist_t, void* stream) {
  if (n < 0 || n_out < 0 || n >= (1LL << 31) || n_out >= (1LL << 31) || !rank_out) return CRB_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  RowsArgs a{};
  a.coords = coords; a.n = (int)n; a.K = ksize[0] * ksize[1] * ksize[2]; a.mode = 2;
  a.s = shape_of(out_shape_dhw);
  a.g = ConvGeom{ksize[0], ksize[1], ksize[2], stride[0], stride[1], stride[2], padding[0], padding[1], padding[2]};
  a.rank = (const uint2*)rank_out; a.n_lookup = (int)n_out;
  a.nbr = nbr_t; a.scatter = nbr; a.mask = mask_t; a.hist = hist_t;
  if (a.K > 32) return CRB_ERR_UNSUPPORTED;
  if (n_out > 0) CRB_HIP(hipMemsetAsync(nbr, 0xff, (size_t)n_out * a.K * 4, st));
  return launch_rows(a, st);
}

extern "C" int crb_table_masks(const int32_t* nbr, int64_t n, int K, uint32_t* mask, int32_t* hist, void* stream) {
  if (n < 0 || n >= (1LL << 31)) return CRB_ERR_ARG;
  RowsArgs a{};
  a.n = (int)n; a.K = K; a.mode = 3; a.nbr = const_cast<int32_t*>(nbr); a.mask = mask; a.hist = hist;
  return launch_rows(a, (hipStream_t)stream);
}

#ifdef CRB_MEASURE
extern "C" int crb_tables_set_skip(int bits) { g_chunk_skip = bits & 7; return CRB_OK; }
#endif

extern "C" int crb_tables_finish(const CrbTablePlan* tables, int n_tables, void* stream) {
  if (n_tables < 0) return CRB_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  for (int t0 = 0; t0 < n_tables; t0 += MAX_TABLES) {
    FinishArgs fa;
    fa.n_tables = 0;
    fa.skip = g_chunk_skip;
    int wgs = 0, blocks = 0;
    for (int t = t0; t < n_tables && fa.n_tables < MAX_TABLES; ++t) {
      const CrbTablePlan& T = tables[t];
      if (T.n < 0 || T.n >= (1LL << 31) || T.K <= 0 || T.K > 32) return CRB_ERR_ARG;
      if (T.n == 0) {
        CRB_HIP(hipMemsetAsync(T.cbase, 0, sizeof(int), st));
        if (T.pair_start) CRB_HIP(hipMemsetAsync(T.pair_start, 0, sizeof(int) * (T.K + 1), st));
        continue;
      }
      if (!T.nbr || !T.mask || !T.hist || !T.perm || !T.cmask || !T.cbase || !T.packed || !T.tile_weight || !T.tile_order)
        return CRB_ERR_ARG;
      if ((T.pair_in || T.pair_out || T.pair_start || T.pair_unit_base) &&
          !(T.pair_in && T.pair_out && T.pair_start && T.pair_unit_base)) return CRB_ERR_ARG;
      fa.t[fa.n_tables] = T;
      fa.chunk_start[fa.n_tables] = wgs;
      fa.block_start[fa.n_tables] = blocks;
      wgs += crb_cdiv(T.n, CHUNK);
      blocks += crb_cdiv(T.n, 256);
      ++fa.n_tables;
    }
    if (fa.n_tables == 0) continue;
    for (int k = fa.n_tables; k <= MAX_TABLES; ++k) { fa.chunk_start[k] = wgs; fa.block_start[k] = blocks; }
    hipLaunchKernelGGL(tables_sort_kernel, dim3(wgs), dim3(1024), 0, st, fa);
    hipLaunchKernelGGL(tables_fill_kernel, dim3(blocks), dim3(256), 0, st, fa);
    hipLaunchKernelGGL(tables_order_kernel, dim3(fa.n_tables * LPT_RANGES), dim3(1024), 0, st, fa);
  }
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}
