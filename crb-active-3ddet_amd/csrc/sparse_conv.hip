// Sparse 3D convolution arithmetic for gfx950 (row a4 of SURVEY §8): output-stationary gather-GEMM
// forward / dgrad on the f32 MFMA (v_mfma_f32_16x16x4_f32, exact f32), pair-list wgrad.
//
// Replaces the gather -> GEMM -> scatter-add inside spconv.pytorch SubMConv3d / SparseConv3d (third-party
// spconv-cu113 v2.1.21) as instantiated by pcdet/models/backbones_3d/spconv_backbone.py:77-117,148-157.
//
// Forward:  Y[i,:] = sum_o X[nbr[i][o],:] @ W[o]          W is (K, Cin, Cout) f32, contiguous
// dgrad  :  the same kernel on the transposed table with W[o]^T (host passes both)
// wgrad  :  dW[o] = sum_{pairs p of offset o} X[pin[p],:]^T (x) dY[pout[p],:]
//
// Tile shape: one 256-thread workgroup = 4 waves x 16 output rows. Lane l of a wave owns A row i = l&15 and
// k-group g = l>>4 of the 16x16x4 MFMA. A lane reads a float4 of its gathered row (16 s + 4 g .. +3), i.e. the
// k index is permuted (k' = 16 s + 4 g + t for MFMA t) — the same permutation indexes W in LDS, so the sum is
// unchanged up to f32 summation order. W[o] is staged in LDS once per workgroup and offset (row stride Cout+4
// keeps the two 16-lane halves of a ds_read_b32 group on different banks).
// Offsets for which no row of the workgroup (resp. wave) has a neighbour are skipped.
// blockIdx is remapped so that consecutive row tiles land on the same XCD (private L2 per XCD).
#include "crb_common.h"
#include "../../include/crb_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef CRB_MEASURE
static unsigned long long* g_wgrad_dbg = nullptr;   // measurement runs: device buffer of 4 u64 per workgroup (see wgrad2)
#else
static constexpr unsigned long long* g_wgrad_dbg = nullptr;
#endif
CRB_KNOB g_wgrad_mode = 0;      // measurement builds of the 64x64 v3 wgrad: 1 = no MFMAs, 2 = no gather pipeline (wrong results)
CRB_KNOB g_wgrad_v1 = 0;        // measurement knob: 0 = default (v3 / v2 by shape), 1 = v1 (16x16x4, register gather) everywhere, 2 = v2 where it exists

namespace {

__device__ __forceinline__ int xcd_remap(int b, int nblocks) {
  // hardware round-robins block b to XCD b%8; give every XCD a contiguous chunk of tiles
  const int per = (nblocks + 7) >> 3;
  int t = (b & 7) * per + (b >> 3);
  return t;
}

// A-operand fragment of one gathered row for all k-steps of a lane
template <int CIN>
struct AFrag {
  static constexpr int NV = CIN >= 16 ? CIN / 16 : (CIN + 3) / 4;
  f32x4 v[NV];     // CIN >= 16: float4 per 16-channel step; CIN < 16: one scalar per k-step in .x
};

template <int CIN>
__device__ __forceinline__ void load_afrag(AFrag<CIN>& a, const float* __restrict__ X, int r, int g) {
  if constexpr (CIN >= 16) {
#pragma unroll
    for (int s = 0; s < CIN / 16; ++s) {
      a.v[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (r >= 0) a.v[s] = *reinterpret_cast<const f32x4*>(X + (int64_t)r * CIN + 16 * s + 4 * g);
    }
  } else {
#pragma unroll
    for (int s = 0; s < (CIN + 3) / 4; ++s) {
      const int k = 4 * s + g;
      a.v[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (r >= 0 && k < CIN) a.v[s][0] = X[(int64_t)r * CIN + k];
    }
  }
}

// rows of a tile are taken through `perm` (rows sorted by their neighbour bit-mask inside chunks, so that the 16 rows
// of a sub-tile share the same set of active offsets: the dense-over-offsets MFMA waste drops from ~55-70 % to ~15-20 %);
// perm == nullptr means identity.
// A workgroup covers 64*SUBT rows: wave w owns SUBT sub-tiles of 16 rows. One staging of W[o] (double buffered in LDS:
// fetched global->registers before the MFMA block, written to the other buffer after it, one barrier per active offset)
// now serves 64*SUBT rows — at SUBT=1 the W traffic through the vector-memory pipe (12 active offsets x 16 KiB per 64
// rows at C=64) exceeded the row gathers themselves. The next (offset, sub-tile)'s gathered rows are prefetched into
// registers while the current one runs on the MFMA pipe.
template <int CIN, int COUT, int SUBT>
__global__ __launch_bounds__(256) void sparse_conv_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                              const int* __restrict__ nbr, const int* __restrict__ perm,
                                                              float* __restrict__ Y, int n_out, int K, int ntiles) {
  constexpr int NB = (COUT + 15) / 16;   // 16-wide output column blocks (last one masked when COUT % 16)
  // LDS layout of W[o]: [k][li][nb] (column nb*16+li stored at li*NB+nb), row stride NB*16 floats, unpadded: a lane's
  // NB column-block values of one k are contiguous -> one ds_read_b128 (b64/b32 for NB=2/1) instead of NB ds_read_b32,
  // and with a 256-B row the 4 k-groups of a wave (rows k, k+4, ...) land on disjoint 16-B slots: conflict free.
  constexpr int WS = NB * 16;
  constexpr int KSTEPS = (CIN + 3) / 4;  // MFMA k-steps
  constexpr int CINP = KSTEPS * 4;
  constexpr int WPT = ((CINP * 16 + 255) / 256) * NB;   // per thread: its (k, li) pairs x all NB column blocks
  constexpr int TROWS = 64 * SUBT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* w_lds0 = reinterpret_cast<float*>(smem);
  float* w_lds1 = w_lds0 + CINP * WS;
  int* nbr_lds = reinterpret_cast<int*>(smem + 2 * sizeof(float) * CINP * WS);  // TROWS * K ints
  __shared__ unsigned wg_mask_sh[4];

  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  if (tile >= ntiles) return;
  const int row0 = tile * TROWS;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int li = lane & 15;
  const int g = lane >> 4;

  const int rows_here = min(TROWS, n_out - row0);
  for (int t = threadIdx.x; t < TROWS * K; t += 256) {
    int r = t / K;
    nbr_lds[t] = (r < rows_here) ? nbr[(int64_t)row0 * K + t] : -1;
  }
  __syncthreads();
  // per sub-tile offset masks (K <= 32); sub-tile u of wave w holds tile rows (w*SUBT + u)*16 ..
  unsigned sm[SUBT];
  unsigned wmask = 0;
#pragma unroll
  for (int u = 0; u < SUBT; ++u) {
    const int myrow = (wave * SUBT + u) * 16 + li;
    unsigned m = 0;
    for (int o = 0; o < K; ++o) {
      bool has = nbr_lds[myrow * K + o] >= 0;
      if (__ballot(has)) m |= (1u << o);
    }
    sm[u] = m;
    wmask |= m;
  }
  if (lane == 0) wg_mask_sh[wave] = wmask;
  __syncthreads();
  unsigned todo = wg_mask_sh[0] | wg_mask_sh[1] | wg_mask_sh[2] | wg_mask_sh[3];

  f32x4 acc[SUBT][NB];
#pragma unroll
  for (int u = 0; u < SUBT; ++u)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[u][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // W staging helpers: element e = (k, li, nb) in LDS order; consecutive threads take consecutive li for a fixed nb so
  // the global reads of a wave stay contiguous 64-B runs and each thread's NB values of one (k, li) are adjacent
  float wreg[WPT];
  auto w_fetch = [&](int o) {
    const float* wsrc = W + (int64_t)o * CIN * COUT;
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
      const int q = (threadIdx.x + 256 * (j / NB));          // (k, li) pair index
      const int nb = j % NB;
      const int k = q >> 4, l = q & 15, c = nb * 16 + l;
      wreg[j] = (q < CINP * 16 && k < CIN && c < COUT) ? wsrc[k * COUT + c] : 0.f;
    }
  };
  auto w_store = [&](float* dst) {
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
      const int q = (threadIdx.x + 256 * (j / NB));
      const int nb = j % NB;
      if (q < CINP * 16) dst[(q >> 4) * WS + (q & 15) * NB + nb] = wreg[j];
    }
  };

  // this wave's work list is the (offset, sub-tile) pairs with sm[u] bit o set, walked offset-major
  auto next_pair = [&](int& o, int& u) {       // advance to the next active pair after (o,u); o = K when exhausted
    while (o < K) {
      ++u;
      if (u >= SUBT) { u = 0; ++o; if (o >= K) return; }
      bool act = false;
#pragma unroll
      for (int v = 0; v < SUBT; ++v) act |= (v == u) && ((sm[v] >> o) & 1u);
      if (act) return;
    }
  };
  auto row_of = [&](int o, int u) -> int {
    return (o < K) ? nbr_lds[((wave * SUBT + u) * 16 + li) * K + o] : -1;
  };

  int cur = todo ? __ffs(todo) - 1 : -1;
  if (cur >= 0) { w_fetch(cur); w_store(w_lds0); }
  int po = -1, pu = SUBT - 1;                 // prefetch cursor
  next_pair(po, pu);
  AFrag<CIN> a_cur, a_nxt;
  load_afrag<CIN>(a_cur, X, row_of(po, pu), g);
  __syncthreads();
  int buf = 0;
  while (cur >= 0) {
    todo &= todo - 1;
    const int nxt = todo ? __ffs(todo) - 1 : -1;
    if (nxt >= 0) w_fetch(nxt);                         // global -> registers, lands during the MFMA block
    const float* wl = buf ? w_lds1 : w_lds0;
#pragma unroll
    for (int u = 0; u < SUBT; ++u) {
      if (!((sm[u] >> cur) & 1u)) continue;             // wave-uniform
      // (po,pu) == (cur,u) here by construction; fetch the following pair's rows before the MFMAs
      next_pair(po, pu);
      load_afrag<CIN>(a_nxt, X, row_of(po, pu), g);
      if constexpr (CIN >= 16) {
#pragma unroll
        for (int s = 0; s < CIN / 16; ++s) {
          float b[4][NB];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float* src = wl + (16 * s + 4 * g + t) * WS + li * NB;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) b[t][nb] = src[nb];
          }
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              acc[u][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur.v[s][t], b[t][nb], acc[u][nb], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
          const float* src = wl + (4 * s + g) * WS + li * NB;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[u][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur.v[s][0], src[nb], acc[u][nb], 0, 0, 0);
        }
      }
      a_cur = a_nxt;
    }
    if (nxt >= 0) w_store(buf ? w_lds0 : w_lds1);
    __syncthreads();
    buf ^= 1;
    cur = nxt;
  }

  // C/D layout of 16x16: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
  for (int u = 0; u < SUBT; ++u)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int srow = row0 + (wave * SUBT + u) * 16 + g * 4 + rg;
      if (srow < n_out) {
        const int row = perm ? perm[srow] : srow;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          if (nb * 16 + li < COUT) Y[(int64_t)row * COUT + nb * 16 + li] = acc[u][nb][rg];
      }
    }
}

// ---------------------------------------------------------------------------------------------------
// v2 of the forward kernel for CIN % 16 == 0 (one 16-row tile per wave). Same tiling, LDS layout and arithmetic order as
// above, restructured after reading the ISA of v1 (tools/pmc_sparse_conv.sh: 39 % of wave cycles parked in s_waitcnt,
// MFMA pipe 51 % busy): the `a_cur = a_nxt` register copies made the compiler wait for the just-issued gather in the
// middle of every MFMA block. Here the two A fragments ping-pong (two copies of the MFMA block, wave-uniform branch), so a
// prefetched row is first touched one phase later, and the W operands of k-group s+1 are read from LDS under the MFMAs of
// group s. Measured on the SECOND bs=16 geometry (tools/bench_sparse_conv.py, v1 -> v2): 97.5 -> 85.7 us at C=32, 218 ->
// 203 us (L3) / 167 -> 153 us (L4) at C=64; with all 27 neighbours present 436 -> 363 us = 117 TF (75 % of the f32 MFMA
// peak). A third variant that split the OUTPUT COLUMNS over the 4 waves (identical work per wave, A tile shared through
// LDS, 2 barriers per phase) was correct but slower (238 us L3, 95 TF dense) and is not kept. What remains on real tables
// is per-phase latency: time fits 25 us + 12.6 us x (offsets in the workgroup union), i.e. a phase costs the same whether 4
// or 1 of its tiles are active — the W[o] global->LDS hand-off and its barrier set a floor the MFMA work does not fill.
// ---------------------------------------------------------------------------------------------------
// cycle accounting of the measurement build (TIMING): [0] waves, [1] total, [2] prologue, [3] load issue, [4] MFMA block,
// [5] W store (incl. its vmcnt wait), [6] barrier wait, [7] epilogue, [8] phases, [9] phases with MFMA work,
// [10] W fetch issue, [11] row-index LDS read ([3] is then the gather issue only), [12] wait at phase start until the
// gather prefetched one phase earlier has landed (explicit vmcnt(0), measurement build only), [13] prologue part 1: table
// copy + perm loads up to the first barrier, [14] part 2: masks up to the second barrier ([2] is then part 3: first W[o]
// fetch + store + first gather issue + barrier)
__device__ unsigned long long g_fwd2_timing[16];

// COMPACT: the neighbour table arrives as one 27-bit mask per row + the present indices packed row after row (cbase =
// exclusive prefix of the masks' popcounts): 8 + 4 P/N bytes per row instead of 4 K (22 against 108 at the 16-channel level,
// where the table was 40 % of the launch's HBM bytes). A row's neighbour through offset o is
// packed[cbase[row] + popcount(mask & ((1 << o) - 1))] when bit o is set.
// Optional epilogue of the forward kernel (inference): y = relu(gamma * ((acc + bias - mean) * rsqrt(var + eps)) + beta) per
// output channel — the conv bias, the BatchNorm1d of the running statistics and the ReLU that follow a sparse conv in the
// reference's post_act_block, in the arithmetic order of bn_apply_kernel (crb_bn_relu_apply). mean == nullptr: plain conv.
struct ConvEpilogue {
  const float* bias;
  const float* gamma;
  const float* beta;
  const float* mean;
  const float* var;
  float eps;
  int relu;
};

template <int CIN, int COUT, bool NOMFMA = false, bool REMAP = true, bool TIMING = false, bool COMPACT = false,
          bool ROWC = false>
__global__ __launch_bounds__(256, 4) void sparse_conv_fwd2_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                               const int* __restrict__ nbr, const int* __restrict__ perm,
                                                               float* __restrict__ Y, int n_out, int K, int ntiles,
                                                               const unsigned* __restrict__ cmask = nullptr,
                                                               const int* __restrict__ cbase = nullptr,
                                                               ConvEpilogue ep = ConvEpilogue{nullptr, nullptr, nullptr, nullptr,
                                                                                              nullptr, 0.f, 0},
                                                               const int* __restrict__ tile_order = nullptr) {
  static_assert(CIN % 16 == 0 && COUT % 16 == 0, "v2 needs whole float4 k-groups and unmasked column blocks");
  static_assert(!ROWC || CIN == 64, "row-contiguous gathers: a lane group owns 64 B of a 256-B row");
  constexpr int NB = (COUT + 15) / 16;
  constexpr int WS = NB * 16;
  constexpr int KS = CIN / 16;                      // k-groups of 16 channels (4 MFMA k-steps each)
  constexpr int WELEMS = CIN * NB * 16;
  constexpr int WPT = (WELEMS + 255) / 256;
  constexpr bool BDB = NB <= 2;                     // double-buffer the B operands while they fit in registers
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* w_lds0 = reinterpret_cast<float*>(smem);
  float* w_lds1 = w_lds0 + CIN * WS;
  int* nbr_lds = reinterpret_cast<int*>(smem + 2 * sizeof(float) * CIN * WS);   // 64 * K ints
  __shared__ unsigned wg_mask_sh[4];
  __shared__ unsigned row_mask_sh[COMPACT ? 64 : 1];
  __shared__ int row_base_sh[COMPACT ? 64 : 1];

  const int pos = REMAP ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  if (pos >= ntiles) return;
  // heaviest-first dispatch inside the XCD's tile range as an indirection (crb_tables_finish): nothing is permuted in memory
  const int tile = tile_order ? tile_order[pos] : pos;
  unsigned long long tk[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_begin = TIMING ? __builtin_readcyclecounter() : 0ULL;
  unsigned long long t_mark = t_begin;
  auto lap = [&](int slot) {
    if constexpr (TIMING) {
      const unsigned long long now = __builtin_readcyclecounter();
      tk[slot] += now - t_mark;
      t_mark = now;
    }
  };
  const int row0 = tile * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  // Prologue and epilogue run in the issue shadows of the other resident workgroups' MFMAs (s_memtime accounting,
  // tools/time_gather_gemm_regions.py: 31 % + 21 % of a wave's lifetime on the real tables), so they are kept short in
  // INSTRUCTIONS: the 64 x K table is copied without a division per element, a wave derives its tile's offset mask from
  // ceil(K/4) LDS reads per lane + an OR butterfly (not K reads + K ballots), and the output-row indirection perm[] is
  // fetched here, long before the final stores need it.
  if constexpr (COMPACT) {
    const int nrow = min(64, n_out - row0);
    const int b0 = cbase[row0], b1 = cbase[row0 + nrow];              // wave-uniform
    if (threadIdx.x < 64) {
      const int t = threadIdx.x;
      row_mask_sh[t] = t < nrow ? cmask[row0 + t] : 0u;
      row_base_sh[t] = (t < nrow ? cbase[row0 + t] : b1) - b0;
    }
    for (int t = threadIdx.x; t < b1 - b0; t += 256) nbr_lds[t] = nbr[b0 + t];      // nbr = the packed index array
  } else {
    const int lim = min(64, n_out - row0) * K;
    for (int t = threadIdx.x; t < 64 * K; t += 256) nbr_lds[t] = (t < lim) ? nbr[(int64_t)row0 * K + t] : -1;
  }
  int out_row[4];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const int srow = row0 + wave * 16 + g * 4 + rg;
    out_row[rg] = srow < n_out ? (perm ? perm[srow] : srow) : -1;
  }
  __syncthreads();
  lap(13);
  const int* my_nbr = nbr_lds + (wave * 16 + li) * K;
  const unsigned my_mask = COMPACT ? row_mask_sh[wave * 16 + li] : 0u;
  const int my_lb = COMPACT ? row_base_sh[wave * 16 + li] : 0;
  auto nbr_of = [&](int o) -> int {
    if constexpr (COMPACT) {
      const int idx = nbr_lds[my_lb + __popc(my_mask & ((1u << o) - 1u))];   // in-bounds also when bit o is clear
      return ((my_mask >> o) & 1u) ? idx : -1;
    } else {
      return my_nbr[o];
    }
  };
  unsigned sm = 0;
  if constexpr (COMPACT) {
    sm = my_mask;
  } else {
    for (int o = g; o < K; o += 4) sm |= (my_nbr[o] >= 0 ? 1u : 0u) << o;      // lane (li, g): offsets g, g+4, ...
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) sm |= (unsigned)__shfl_xor((int)sm, d);
  sm = __builtin_amdgcn_readfirstlane(sm);
  if (lane == 0) wg_mask_sh[wave] = sm;
  __syncthreads();
  lap(14);
  unsigned todo = wg_mask_sh[0] | wg_mask_sh[1] | wg_mask_sh[2] | wg_mask_sh[3];

  f32x4 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float wreg[WPT];
  auto w_fetch = [&](int o) {
    const float* wsrc = W + (int64_t)o * CIN * COUT;
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
      const int q = threadIdx.x + 256 * (j / NB);
      const int nb = j % NB;
      const int k = q >> 4, l = q & 15, c = nb * 16 + l;
      wreg[j] = wsrc[k * COUT + c];                  // CIN*16 is a multiple of 256: no tail predicate
    }
  };
  auto w_store = [&](float* dst) {
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
      const int q = threadIdx.x + 256 * (j / NB);
      const int nb = j % NB;
      dst[(q >> 4) * WS + (q & 15) * NB + nb] = wreg[j];
    }
  };
  // Loads are issued unconditionally (invalid rows read row 0 and are zeroed by a select at use): every path through a
  // phase then issues the same number of VMEM loads, so the compiler's in-order vmcnt bookkeeping can leave the prefetch
  // outstanding across the phase instead of draining it at the next control-flow join.
  // ROWC (CIN = 64): the four lanes of a quad (rows 4q..4q+3 of the tile, same lane group g) read the SAME row per
  // instruction — 64 contiguous bytes, channels 16g..16g+15 of row 4q+i in instruction i — instead of four rows at one
  // channel offset: a quarter of the cache-line look-ups per instruction. Lane j then holds piece j of the four rows and
  // needs the four pieces of row j: a 4x4 transpose of 16-byte elements inside the quad, done on the fly in the MFMA block
  // (two DPP exchange stages per dword). The channel a lane group feeds to MFMA step (s, t) becomes 16g + 4s + t (it was
  // 16s + 4g + t); B is read from the matching W row, so the products and their order per output element are unchanged.
  auto load_a = [&](f32x4 (&a)[KS], const float* xbase, int r) {
    if constexpr (ROWC) {
      const int rr = r < 0 ? 0 : r;
      const int j = lane & 3;
      const int r0q = __builtin_amdgcn_mov_dpp(rr, 0x00, 0xf, 0xf, true);      // quad_perm [i,i,i,i]: row of lane 4q+i
      const int r1q = __builtin_amdgcn_mov_dpp(rr, 0x55, 0xf, 0xf, true);
      const int r2q = __builtin_amdgcn_mov_dpp(rr, 0xAA, 0xf, 0xf, true);
      const int r3q = __builtin_amdgcn_mov_dpp(rr, 0xFF, 0xf, 0xf, true);
      const float* base = xbase + 16 * g + 4 * j;
      a[0] = *reinterpret_cast<const f32x4*>(base + (int64_t)r0q * CIN);
      a[1] = *reinterpret_cast<const f32x4*>(base + (int64_t)r1q * CIN);
      a[2] = *reinterpret_cast<const f32x4*>(base + (int64_t)r2q * CIN);
      a[3] = *reinterpret_cast<const f32x4*>(base + (int64_t)r3q * CIN);
    } else {
      const f32x4* src = reinterpret_cast<const f32x4*>(xbase + (int64_t)(r < 0 ? 0 : r) * CIN + 4 * g);
#pragma unroll
      for (int s = 0; s < KS; ++s) a[s] = src[4 * s];
    }
  };
  auto load_b = [&](float (&b)[4][NB], const float* wl, int s) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float* src = wl + (16 * s + 4 * g + t) * WS + li * NB;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) b[t][nb] = src[nb];
    }
  };
  // dword t of the four raw pieces v[0..3] (piece i = row 4q+i) -> m[s] = dword t of piece s of THIS lane's row
  auto quad_transpose = [&](const f32x4 (&v)[KS], int t, float (&m)[4]) {
    const bool odd = lane & 1, hi = lane & 2;
    auto x1 = [](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true)); };
    auto x2 = [](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true)); };
    // every exchange is evaluated by ALL lanes before the selects (a DPP read under a lane-dependent branch would see its
    // source lanes switched off)
    constexpr int i1 = KS > 1 ? 1 : 0, i2 = KS > 2 ? 2 : 0, i3 = KS > 3 ? 3 : 0;      // (only instantiated paths with KS = 4 call this)
    const float p0 = x1(v[0][t]), p1 = x1(v[i1][t]), p2 = x1(v[i2][t]), p3 = x1(v[i3][t]);
    const float n0 = odd ? p1 : v[0][t];
    const float n1 = odd ? v[i1][t] : p0;
    const float n2 = odd ? p3 : v[i2][t];
    const float n3 = odd ? v[i3][t] : p2;
    const float q0 = x2(n0), q1 = x2(n1), q2 = x2(n2), q3 = x2(n3);
    m[0] = hi ? q2 : n0;
    m[1] = hi ? q3 : n1;
    m[2] = hi ? n2 : q0;
    m[3] = hi ? n3 : q1;
  };
  auto mfma_block = [&](const f32x4 (&a)[KS], bool valid, const float* wl) {
    if constexpr (ROWC && !NOMFMA) {
      // t-major: the exchange of dword t+1 and the B reads of step t+1 have no dependence on the 16 MFMAs of step t
      float m[2][4];
      float b[2][4][NB];
      auto load_bt = [&](float (&bb)[4][NB], int t) {
#pragma unroll
        for (int sgrp = 0; sgrp < 4; ++sgrp) {
          const float* src = wl + (16 * g + 4 * sgrp + t) * WS + li * NB;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) bb[sgrp][nb] = src[nb];
        }
      };
      quad_transpose(a, 0, m[0]);
      load_bt(b[0], 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t + 1 < 4) {
          quad_transpose(a, t + 1, m[(t + 1) & 1]);
          load_bt(b[(t + 1) & 1], t + 1);
        }
#pragma unroll
        for (int sgrp = 0; sgrp < 4; ++sgrp) {
          const float av = valid ? m[t & 1][sgrp] : 0.f;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[t & 1][sgrp][nb], acc[nb], 0, 0, 0);
        }
      }
    } else if constexpr (NOMFMA) {                          // measurement build: staging chain only (result is garbage)
#pragma unroll
      for (int s = 0; s < KS; ++s) acc[0][0] += valid ? a[s][0] + wl[(16 * s + 4 * g) * WS + li * NB] : 0.f;
    } else if constexpr (BDB) {
      float b[2][4][NB];
      load_b(b[0], wl, 0);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s + 1 < KS) load_b(b[(s + 1) & 1], wl, s + 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float av = valid ? a[s][t] : 0.f;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[s & 1][t][nb], acc[nb], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        float b[4][NB];
        load_b(b, wl, s);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float av = valid ? a[s][t] : 0.f;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[t][nb], acc[nb], 0, 0, 0);
        }
      }
    }
  };

  int cur = todo ? __ffs(todo) - 1 : -1;
  if (cur >= 0) { w_fetch(cur); w_store(w_lds0); }
  // Phases are unrolled by two so the two A fragments alternate statically: phase `cur` multiplies fragment a0 while the
  // rows of the workgroup's NEXT offset are fetched into a1, then the roles swap. Every wave prefetches for every phase of
  // the workgroup (also for offsets its own tile lacks: those rows are -1 -> row 0, never multiplied), so all paths issue
  // the same loads and only the MFMA block itself sits under the wave-uniform `has` branch.
  f32x4 a0[KS], a1[KS];
  int r0 = cur >= 0 ? nbr_of(cur) : -1, r1 = -1;
  load_a(a0, X, r0);
  __syncthreads();
  lap(2);
  while (cur >= 0) {
    {                                                // even phase: W in w_lds0, A in a0
      todo &= todo - 1;
      const int nxt = todo ? __ffs(todo) - 1 : -1;
      const int oq = nxt >= 0 ? nxt : cur;           // the last phase re-fetches its own offset (unused)
      if constexpr (TIMING) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); lap(12); }
      w_fetch(oq);
      lap(10);
      r1 = nbr_of(oq);
      if constexpr (TIMING) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); lap(11); }
      load_a(a1, X, r1);
      lap(3);
      if ((sm >> cur) & 1u) { mfma_block(a0, r0 >= 0, w_lds0); tk[9] += 1; }
      lap(4);
      w_store(w_lds1);
      lap(5);
      __syncthreads();
      lap(6);
      tk[8] += 1;
      cur = nxt;
    }
    if (cur < 0) break;
    {                                                // odd phase: W in w_lds1, A in a1
      todo &= todo - 1;
      const int nxt = todo ? __ffs(todo) - 1 : -1;
      const int oq = nxt >= 0 ? nxt : cur;
      if constexpr (TIMING) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); lap(12); }
      w_fetch(oq);
      lap(10);
      r0 = nbr_of(oq);
      if constexpr (TIMING) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); lap(11); }
      load_a(a0, X, r0);
      lap(3);
      if ((sm >> cur) & 1u) { mfma_block(a1, r1 >= 0, w_lds1); tk[9] += 1; }
      lap(4);
      w_store(w_lds0);
      lap(5);
      __syncthreads();
      lap(6);
      tk[8] += 1;
      cur = nxt;
    }
  }

  if (ep.mean != nullptr) {                          // wave-uniform
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int c = nb * 16 + li;
      const float mu = ep.mean[c], is = rsqrtf(ep.var[c] + ep.eps), ga = ep.gamma[c], be = ep.beta[c];
      const float bi = ep.bias ? ep.bias[c] : 0.f;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        float v = acc[nb][rg];
        if (ep.bias) v = v + bi;
        v = ga * ((v - mu) * is) + be;
        acc[nb][rg] = (ep.relu && !(v > 0.f)) ? 0.f : v;
      }
    }
  }
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    if (out_row[rg] >= 0) {
      float* dst = Y + (int64_t)out_row[rg] * COUT + li;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) dst[nb * 16] = acc[nb][rg];
    }
  }
  if constexpr (TIMING) {
    lap(7);
    if (lane == 0) {
      atomicAdd(&g_fwd2_timing[0], 1ULL);
      atomicAdd(&g_fwd2_timing[1], __builtin_readcyclecounter() - t_begin);
      for (int k = 2; k < 15; ++k) atomicAdd(&g_fwd2_timing[k], tk[k]);
    }
  }
}

// ---- low-channel forward (round 4): C_in in {4, 16}, C_out = 16 on the compact table ------------------------------------------
// The phase kernel above pays per kernel offset present in a 64-row tile: a 1 KB W[o] hand-over through LDS, a barrier, a
// dependent gather - at 3.5 neighbours per row that is ~10 dependent phases of 4 MFMAs per tile: latency-bound (17 / 25 us at
// level 1 against 4 - 6 us for a copy of the same bytes, profiles/r04_lowchannel_floor.txt). Here all K weight matrices (27.6 KB
// at 16 x 16) are either read from L1 / L2 where they stay hot (WLDS = false: 4 waves per workgroup, one tile per wave, no barrier
// at all - the product instance) or RESIDENT in LDS in MFMA B-operand order for a 16-wave workgroup (WLDS = true, measurement
// builds). A wave owns 16-row tiles end to end and walks only the offsets present in ITS 16 rows (wave-uniform OR of the masks),
// NB offsets per round: index reads from the wave's LDS copy of the tile's packed indices, gathers (16 bytes per lane =
// channels 4 kq .. 4 kq + 3 of the row), W[o], 4 MFMAs (1 MFMA at C_in = 4).
// Measured (tools/lowchannel_floor.py): 4 -> 16 20.1 us (v1 kernel on the (N,K) table 25.4), 16 -> 16 28.5 us (phase kernel
// 17.2): three dependent round trips per tile (masks / bases -> packed indices -> rows) bound it, not the weights; the product
// uses it for C_in = 4 only.
// Products and their order per output element are those of sparse_conv_fwd2_kernel (offsets ascending, channel 4 kq + t at
// MFMA step t): results are bit-identical to it.
template <int CIN, int NB, bool WLDS>
__global__ __launch_bounds__(WLDS ? 1024 : 256) void sparse_conv_fwd_lc_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                                  const unsigned* __restrict__ cmask,
                                                                  const int* __restrict__ cbase, const int* __restrict__ packed,
                                                                  const int* __restrict__ perm, float* __restrict__ Y, int n_out,
                                                                  int K, int ntiles, ConvEpilogue ep) {
  static_assert(CIN == 4 || CIN == 16, "low-channel instance");
  constexpr int COUT = 16, WO = CIN * COUT;     // floats per offset
  constexpr int TILE_INTS = 16 * 32 + 16;       // packed indices of 16 rows (<= 16 K) + the 16 output rows
  extern __shared__ __attribute__((aligned(16))) float lc_lds[];
  float* const Wl = lc_lds;
  const int T = threadIdx.x, lane = T & 63, wave = __builtin_amdgcn_readfirstlane(T >> 6), nw = blockDim.x >> 6;
  int* const my = reinterpret_cast<int*>(lc_lds + (WLDS ? ((K * WO + 3) & ~3) : 0)) + wave * TILE_INTS;
  // weights in B-operand order. C_in = 16: [o][kq][co][t] = W[o][4 kq + t][co]: lane (co, kq) reads its four steps as ONE 16-byte
  // read at lane * 16 bytes (conflict-free); C_in = 4: [o][kq][co] = W[o][kq][co]
  if constexpr (WLDS) {
    for (int e = T; e < K * WO; e += blockDim.x) {
      const int o = e / WO, r = e - o * WO, ci = r >> 4, co = r & 15;
      const int d = CIN == 16 ? o * WO + (((ci >> 2) * 16 + co) << 2) + (ci & 3) : o * WO + ci * 16 + co;
      Wl[d] = W[e];
    }
    __syncthreads();
  }
  const int i = lane & 15, kq = lane >> 4;
  const int t0 = (int)((int64_t)blockIdx.x * ntiles / gridDim.x), t1 = (int)((int64_t)(blockIdx.x + 1) * ntiles / gridDim.x);
  for (int t = t0 + wave; t < t1; t += nw) {
    const int row0 = t * 16, srow = row0 + i;
    const bool live = srow < n_out;
    const unsigned mask = live ? cmask[srow] : 0u;
    const int base = cbase[live ? srow : n_out];
    const int b0 = __builtin_amdgcn_readfirstlane(base);                     // lane 0: row0 < n_out always
    const int b1 = cbase[min(row0 + 16, n_out)];                             // wave-uniform address
    for (int e = lane; e < b1 - b0; e += 64) my[e] = packed[b0 + e];
    if (kq == 0) my[16 * 32 + i] = live ? (perm ? perm[srow] : srow) : -1;
    // offsets present in the tile: OR over the 16 rows (every 16-lane row of the wave holds the same 16 masks)
    unsigned U = mask;
    U |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)U, 0x128, 0xf, 0xf, false);   // row_ror:8
    U |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)U, 0x124, 0xf, 0xf, false);
    U |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)U, 0x122, 0xf, 0xf, false);
    U |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)U, 0x121, 0xf, 0xf, false);
    unsigned Us = (unsigned)__builtin_amdgcn_readfirstlane((int)U);
    const int rel = base - b0;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    // NB offsets per round: their index reads, then their gathers, are in flight TOGETHER (one offset per round made a tile a
    // chain of |U| dependent LDS + L2 round trips: 34 us per launch against 17 for the phase kernel)
    while (Us) {
      int o[NB];
      bool ov[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) {                                         // scalar: the next NB set bits
        ov[j] = Us != 0;
        o[j] = ov[j] ? __builtin_ctz(Us) : 0;
        Us = Us & (Us - 1);
      }
      bool has[NB];
      int idx[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        has[j] = ov[j] && ((mask >> o[j]) & 1u);
        const int rank = __popc(mask & ((1u << o[j]) - 1u));
        idx[j] = my[has[j] ? rel + rank : 0];                                // (the wave's own LDS writes: executed in order)
      }
      if constexpr (CIN == 16) {
        f32x4 av[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) av[j] = *reinterpret_cast<const f32x4*>(X + (int64_t)(has[j] ? idx[j] : 0) * CIN + 4 * kq);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          if (ov[j]) {                                                       // wave-uniform
            f32x4 bv;
            if constexpr (WLDS) bv = *reinterpret_cast<const f32x4*>(Wl + o[j] * WO + lane * 4);
            else {                                                           // W[o][4 kq + t][co] from L1 / L2 (27 KB, hot)
              const float* wp = W + o[j] * WO + (4 * kq) * 16 + i;
              bv = (f32x4){wp[0], wp[16], wp[32], wp[48]};
            }
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(has[j] ? av[j][tt] : 0.f, bv[tt], acc, 0, 0, 0);
          }
        }
      } else {
        float av[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) av[j] = X[(int64_t)(has[j] ? idx[j] : 0) * CIN + kq];
#pragma unroll
        for (int j = 0; j < NB; ++j)
          if (ov[j]) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(has[j] ? av[j] : 0.f, WLDS ? Wl[o[j] * WO + lane] : W[o[j] * WO + lane], acc, 0, 0, 0);
      }
    }
    if (ep.mean != nullptr) {                          // wave-uniform; arithmetic order of bn_apply_kernel (as sparse_conv_fwd2_kernel)
      const float mu = ep.mean[i], is = rsqrtf(ep.var[i] + ep.eps), ga = ep.gamma[i], be = ep.beta[i];
      const float bi = ep.bias ? ep.bias[i] : 0.f;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        float v = acc[rg];
        if (ep.bias) v = v + bi;
        v = ga * ((v - mu) * is) + be;
        acc[rg] = (ep.relu && !(v > 0.f)) ? 0.f : v;
      }
    }
    // accumulator register rg = tile row 4 kq + rg, column = output channel i
    const int4 orow = *reinterpret_cast<const int4*>(my + 16 * 32 + 4 * kq);
    if (orow.x >= 0) Y[(int64_t)orow.x * COUT + i] = acc[0];
    if (orow.y >= 0) Y[(int64_t)orow.y * COUT + i] = acc[1];
    if (orow.z >= 0) Y[(int64_t)orow.z * COUT + i] = acc[2];
    if (orow.w >= 0) Y[(int64_t)orow.w * COUT + i] = acc[3];
  }
}

// one 32-bit neighbour mask per row (bit o set <=> nbr[row][o] >= 0); sort key for the row permutation
__global__ __launch_bounds__(256) void nbr_mask_kernel(const int* __restrict__ nbr, int n, int K, int* __restrict__ mask) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned m = 0;
  for (int o = 0; o < K; ++o) m |= (nbr[(int64_t)i * K + o] >= 0 ? 1u : 0u) << o;
  mask[i] = (int)m;
}

__global__ __launch_bounds__(256) void nbr_permute_kernel(const int* __restrict__ nbr, const int* __restrict__ perm,
                                                          int n, int K, int* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)n * K) return;
  const int i = (int)(t / K), o = (int)(t - (int64_t)i * K);
  out[t] = nbr[(int64_t)perm[i] * K + o];
}

// compact neighbour table (consumed by sparse_conv_fwd2_kernel<.., COMPACT = true>): row i of the kernel order is row
// perm[i] of nbr. Pass 1: cmask[i]; device scan of the popcounts -> cbase; pass 2: half a wave per row, lane o looks at
// offset o (one coalesced 4K-byte read per row) and writes its index at cbase[i] + rank of bit o in the mask.
__global__ __launch_bounds__(256) void nbr_compact_mask_kernel(const int* __restrict__ nbr, const int* __restrict__ perm,
                                                               int n, int K, unsigned* __restrict__ cmask) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int i = t >> 5, o = t & 31;
  const bool live = i < n;
  const int src = live ? (perm ? perm[i] : i) : 0;
  const bool has = live && o < K && nbr[(int64_t)src * K + o] >= 0;
  const unsigned long long b = __ballot(has);
  const unsigned m = (unsigned)(((threadIdx.x & 63) < 32) ? (b & 0xffffffffULL) : (b >> 32));
  if (live && o == 0) cmask[i] = m;
}

__global__ __launch_bounds__(256) void nbr_compact_fill_kernel(const int* __restrict__ nbr, const int* __restrict__ perm,
                                                               int n, int K, const unsigned* __restrict__ cmask,
                                                               const int* __restrict__ cbase, int* __restrict__ packed) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int i = t >> 5, o = t & 31;
  if (i >= n || o >= K) return;
  const unsigned m = cmask[i];
  if (!((m >> o) & 1u)) return;
  const int src = perm ? perm[i] : i;
  packed[cbase[i] + __popc(m & ((1u << o) - 1u))] = nbr[(int64_t)src * K + o];
}

// ---------------------------------------------------------------------------------------------------
// wgrad: grid (K, S). Workgroup (o, s) reduces its slice of offset o's pairs into a (CIN x COUT) partial.
// Waves are arranged (ci-wave, pair-slice): WCI waves along Cin (each CIPW 16-row blocks), 4/WCI pair slices.
// ---------------------------------------------------------------------------------------------------
template <int CIN, int COUT>
struct WgradCfg {
  static constexpr int NCI = (CIN + 15) / 16;
  static constexpr int NB = (COUT + 15) / 16;
  // ci blocks per wave. With the Cin rows split over the 4 waves every wave re-loads the same dY rows (20 loads per 16
  // MFMAs at C = 64); a wave that keeps more of the Cin x Cout tile and takes its own pair slice instead loads each value
  // once but pays in accumulator registers. Measured (L2 / L3 geometry, us): 32x32 CIPW 1 / 2 = 109 / 99;
  // 64x64 CIPW 1 / 2 / 4 = 265 / 289 / 494 (occupancy collapses with 32-64 accumulator VGPRs) -> whole tile only up to 32x32.
  static constexpr int TILES = NCI * NB;
  static constexpr int CIPW = TILES <= 4 ? NCI : (NCI >= 4 ? NCI / 4 : 1);
  static constexpr int WCI = NCI / CIPW;                 // waves along ci (1,2,4)
  static constexpr int SLICES = 4 / WCI;                 // pair slices per workgroup
};

// plan (ints, written by wgrad_plan_kernel): [0] = 16-pair blocks per workgroup, [1] = workgroups in use,
// [2 + o] = first workgroup of offset o (o = 0..K). Every workgroup gets the same number of pairs of ONE offset: the centre
// offset of a SubM conv holds N pairs, a corner offset a few thousand — with a fixed number of splits per offset the centre
// workgroups ran 3x longer than the average and the launch ended on them.
__global__ __launch_bounds__(64) void wgrad_plan_kernel(const int* __restrict__ pstart, int K, int target,
                                                        int* __restrict__ plan, int shift = 4) {
  const int o = threadIdx.x;
  const int nb16 = o < K ? (pstart[o + 1] - pstart[o] + (1 << shift) - 1) >> shift : 0;   // pair blocks of 2^shift pairs
  int total = nb16;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) total += __shfl_xor(total, d);
  int cb = (total + target - 1) / target;
  cb = cb < 1 ? 1 : cb;
  const int nwg = (nb16 + cb - 1) / cb;
  int incl = nwg;                                   // inclusive prefix over lanes
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(incl, d);
    if (o >= d) incl += v;
  }
  if (o <= K) plan[2 + o] = incl - nwg;             // lane K: nwg = 0 -> total
  if (o == 0) plan[0] = cb;
  if (o == K) plan[1] = incl;
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void sparse_conv_wgrad_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                                const int* __restrict__ pin, const int* __restrict__ pout,
                                                                const int* __restrict__ pstart, const int* __restrict__ plan,
                                                                float* __restrict__ partial /* (workgroup,CIN,COUT) */,
                                                                int K) {
  using C = WgradCfg<CIN, COUT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);   // SLICES>1: (SLICES-1) * CINP16 * COUT floats
  // consecutive workgroups (= consecutive pair ranges of one offset, touching neighbouring rows) share an XCD and its L2
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  if (wg >= plan[1]) return;
  const int cb = plan[0];
  int o = 0;
  for (int k = 1; k < K; ++k)
    if (plan[2 + k] <= wg) o = k;                  // starts are non-decreasing; empty offsets share their successor's start
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int wci = wave % C::WCI, slice = wave / C::WCI;
  const int p0 = pstart[o], p1 = pstart[o + 1];
  const int blk_lo = (wg - plan[2 + o]) * cb;
  const int nblocks = min((p1 - p0 + 15) >> 4, blk_lo + cb);

  f32x4 acc[C::CIPW][C::NB];
#pragma unroll
  for (int a = 0; a < C::CIPW; ++a)
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb) acc[a][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Each step covers 16 consecutive pairs (4 MFMA k-steps); steps are dealt round-robin to the pair slices of the
  // workgroup. Software pipeline, statically double-buffered (two copies of the body): pair indices two steps ahead, the
  // gathered X / dY values one step ahead, MFMAs on the current step. All loads are unconditional (clamped address, value
  // zeroed by a select at use) so that every path issues the same number of VMEM loads and nothing drains early.
  struct Idx { int ji[4], io[4]; };
  struct Feat { float av[4][C::CIPW], bv[4][C::NB]; };
  auto load_idx = [&](Idx& x, int b) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int p = p0 + b * 16 + u * 4 + g;
      p = p < p1 ? p : p1 - 1;                     // p1 > p0 for every workgroup in use
      x.ji[u] = pin[p];
      x.io[u] = pout[p];
    }
  };
  auto load_feat = [&](Feat& f, const Idx& x) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int nb = 0; nb < C::NB; ++nb) {
        const int c = nb * 16 + li;
        f.bv[u][nb] = dY[(int64_t)x.io[u] * COUT + (c < COUT ? c : COUT - 1)];
      }
#pragma unroll
      for (int a = 0; a < C::CIPW; ++a) {
        const int ci = (wci * C::CIPW + a) * 16 + li;
        f.av[u][a] = X[(int64_t)x.ji[u] * CIN + (ci < CIN ? ci : CIN - 1)];
      }
    }
  };
  auto mfma_step = [&](const Feat& f, int b) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool v = (b < nblocks) && (p0 + b * 16 + u * 4 + g < p1);
#pragma unroll
      for (int a = 0; a < C::CIPW; ++a) {
        const float av = (v && (wci * C::CIPW + a) * 16 + li < CIN) ? f.av[u][a] : 0.f;
#pragma unroll
        for (int nb = 0; nb < C::NB; ++nb) {
          const float bv = (nb * 16 + li < COUT) ? f.bv[u][nb] : 0.f;
          acc[a][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[a][nb], 0, 0, 0);
        }
      }
    }
  };
  constexpr int BS = C::SLICES;
  int blk = blk_lo + slice;
  Idx i0, i1;
  Feat f0, f1;
  load_idx(i0, blk);
  load_feat(f0, i0);
  load_idx(i1, blk + BS);
  while (blk < nblocks) {
    load_feat(f1, i1);
    load_idx(i0, blk + 2 * BS);
    mfma_step(f0, blk);
    blk += BS;
    if (blk >= nblocks) break;
    load_feat(f0, i0);
    load_idx(i1, blk + 2 * BS);
    mfma_step(f1, blk);
    blk += BS;
  }

  // reduce pair slices through LDS (fixed order), slice 0 writes the partial
  constexpr int CINP16 = C::NCI * 16;
  if constexpr (C::SLICES > 1) {
    if (slice > 0) {
#pragma unroll
      for (int a = 0; a < C::CIPW; ++a)
#pragma unroll
        for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int ci = (wci * C::CIPW + a) * 16 + g * 4 + rg;
            red[((slice - 1) * CINP16 + ci) * (C::NB * 16) + nb * 16 + li] = acc[a][nb][rg];
          }
    }
    __syncthreads();
  }
  if (slice == 0) {
    float* dst = partial + (int64_t)wg * CIN * COUT;
#pragma unroll
    for (int a = 0; a < C::CIPW; ++a)
#pragma unroll
      for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int ci = (wci * C::CIPW + a) * 16 + g * 4 + rg;
          float v = acc[a][nb][rg];
          if constexpr (C::SLICES > 1) {
#pragma unroll
            for (int sl = 1; sl < C::SLICES; ++sl) v += red[((sl - 1) * CINP16 + ci) * (C::NB * 16) + nb * 16 + li];
          }
          if (ci < CIN && nb * 16 + li < COUT) dst[ci * COUT + nb * 16 + li] = v;
        }
  }
}

// ---------------------------------------------------------------------------------------------------
// wgrad v2 (CIN % 32 == 0 and COUT % 32 == 0): the gathered rows of a 32-pair step are staged ONCE per workgroup in LDS
// with 16-byte loads (a 64-float row = 16 lanes x dwordx4: whole 256-B rows, against v1's 64-B segments fetched again by
// every wave that needs them), double-buffered with one barrier per step, and the Cin x Cout tile is cut into 32x32 blocks
// on v_mfma_f32_32x32x2_f32 (pair index on the MFMA k-dim: A = X[pin]^T from LDS as [pair][ci], B = dY[pout] as [pair][co];
// one conflict-free ds_read_b32 per operand, half the LDS reads per flop of the 16x16x4 form). Waves own blocks; when the
// tile has fewer than 4 blocks the waves split the 16 k-steps of a step instead and add their accumulators through LDS in
// a fixed order. Same plan / partial / reduce scheme as v1 (deterministic), in units of 32 pairs.
template <int CIN, int COUT>
struct Wgrad2Cfg {
  static constexpr int NBI = CIN / 32, NBO = COUT / 32, BLOCKS = NBI * NBO;
  // pairs per step: >= 8 MFMAs per wave between two barriers (a 32x32 tile gives each wave only a quarter of the k-steps)
  static constexpr int PS = BLOCKS == 1 ? 64 : 32;
  static constexpr int SLICES = BLOCKS >= 4 ? 1 : 4 / BLOCKS;          // k-step slices per step
  static constexpr int WB = 4 / SLICES;                                // waves along blocks
  static constexpr int BPW = BLOCKS / WB;                              // blocks per wave
  // blocks of a wave: BI_PW x BO_PW sub-grid (share the A read along bo, the B read along bi)
  static constexpr int BO_PW = BPW >= NBO ? NBO : BPW;
  static constexpr int BI_PW = BPW / BO_PW;
  static constexpr int WBO = NBO / BO_PW;                              // waves along bo
  static constexpr int ROWF = CIN + COUT;                              // floats per staged pair
  static constexpr int CHUNKS = PS * ROWF / 4;                         // float4 chunks per step
  static constexpr int CPT = CHUNKS / 256;                             // per thread
  static constexpr size_t LDS_BYTES = 2 * sizeof(float) * PS * ROWF;
  static_assert(CHUNKS % 256 == 0, "staging assumes a whole number of float4 chunks per thread");
  static_assert(BPW * WB == BLOCKS && BI_PW * BO_PW == BPW, "block split");
};

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void sparse_conv_wgrad2_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                                 const int* __restrict__ pin, const int* __restrict__ pout,
                                                                 const int* __restrict__ pstart, const int* __restrict__ plan,
                                                                 float* __restrict__ partial /* (workgroup,CIN,COUT) */,
                                                                 int K, unsigned long long* __restrict__ dbg = nullptr) {
  using C = Wgrad2Cfg<CIN, COUT>;
  const unsigned long long t_start = dbg ? wall_clock64() : 0ULL;   // 100 MHz, common to all XCDs (s_memtime is per XCD)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* stage = reinterpret_cast<float*>(smem);            // [2][PS][CIN + COUT]: X row then dY row of each pair
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  if (wg >= plan[1]) return;
  const int cb = plan[0];
  int o = 0;
  for (int k = 1; k < K; ++k)
    if (plan[2 + k] <= wg) o = k;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p0 = pstart[o], p1 = pstart[o + 1];
  const int blk_lo = (wg - plan[2 + o]) * cb;
  const int nblocks = min((p1 - p0 + C::PS - 1) / C::PS, blk_lo + cb);
  const int wb = wave % C::WB, slice = wave / C::WB;
  const int bi0 = (wb / C::WBO) * C::BI_PW, bo0 = (wb % C::WBO) * C::BO_PW;

  f32x16 acc[C::BI_PW][C::BO_PW];
#pragma unroll
  for (int a = 0; a < C::BI_PW; ++a)
#pragma unroll
    for (int b = 0; b < C::BO_PW; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // staging assignment: chunk c = t + 256 j covers float4 `col` of pair row `pr`; X part first, then dY part
  int c_pr[C::CPT], c_col[C::CPT];
  bool c_isx[C::CPT];
#pragma unroll
  for (int j = 0; j < C::CPT; ++j) {
    const int c = (int)threadIdx.x + 256 * j;
    constexpr int XCH = C::PS * CIN / 4;
    c_isx[j] = c < XCH;
    const int cc = c_isx[j] ? c : c - XCH;
    const int per = c_isx[j] ? CIN / 4 : COUT / 4;
    c_pr[j] = cc / per;
    c_col[j] = cc - c_pr[j] * per;
  }
  struct Idx { int row[C::CPT]; };
  struct Feat { f32x4 v[C::CPT]; };
  auto load_idx = [&](Idx& x, int b) {
#pragma unroll
    for (int j = 0; j < C::CPT; ++j) {
      int p = p0 + b * C::PS + c_pr[j];
      p = p < p1 ? p : p1 - 1;                                  // clamped: value zeroed at the LDS write
      x.row[j] = c_isx[j] ? pin[p] : pout[p];
    }
  };
  auto load_feat = [&](Feat& f, const Idx& x) {
#pragma unroll
    for (int j = 0; j < C::CPT; ++j) {
      const float* src = c_isx[j] ? X + (int64_t)x.row[j] * CIN : dY + (int64_t)x.row[j] * COUT;
      f.v[j] = *reinterpret_cast<const f32x4*>(src + c_col[j] * 4);
    }
  };
  auto store_feat = [&](const Feat& f, int b, float* buf) {
#pragma unroll
    for (int j = 0; j < C::CPT; ++j) {
      const bool ok = (b < nblocks) && (p0 + b * C::PS + c_pr[j] < p1);
      f32x4 v = f.v[j];
      if (!ok) v = (f32x4){0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(buf + c_pr[j] * C::ROWF + (c_isx[j] ? 0 : CIN) + c_col[j] * 4) = v;
    }
  };
  const int l31 = lane & 31, kh = lane >> 5;
  auto compute = [&](const float* buf) {
    constexpr int KSTEPS = C::PS / 2, KPS = KSTEPS / C::SLICES;
    float av[2][C::BI_PW], bv[2][C::BO_PW];
    auto fetch = [&](int q, int s) {               // operands one k-step ahead of the MFMAs (see wgrad v3)
      const float* row = buf + (2 * (slice * KPS + q) + kh) * C::ROWF;
#pragma unroll
      for (int a = 0; a < C::BI_PW; ++a) av[s][a] = row[(bi0 + a) * 32 + l31];
#pragma unroll
      for (int b = 0; b < C::BO_PW; ++b) bv[s][b] = row[CIN + (bo0 + b) * 32 + l31];
    };
    fetch(0, 0);
#pragma unroll
    for (int q = 0; q < KPS; ++q) {
      if (q + 1 < KPS) fetch(q + 1, (q + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < C::BI_PW; ++a)
#pragma unroll
        for (int b = 0; b < C::BO_PW; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q & 1][a], bv[q & 1][b], acc[a][b], 0, 0, 0);
    }
  };

  // pipeline: indices two steps ahead, gathered rows one step ahead (registers), LDS double-buffered, one barrier per step
  Idx ix;
  Feat ft;
  int blk = blk_lo;
  load_idx(ix, blk);
  load_feat(ft, ix);
  load_idx(ix, blk + 1);
  int cur = 0;
  while (blk < nblocks) {
    float* buf = stage + cur * (C::PS * C::ROWF);
    store_feat(ft, blk, buf);
    __syncthreads();
    if (blk + 1 < nblocks) {
      load_feat(ft, ix);
      load_idx(ix, blk + 2);
    }
    compute(buf);
    cur ^= 1;
    ++blk;
  }

  // slices > 1: add the slices' accumulators through LDS in a fixed order (reuses the staging buffers)
  float* dst = partial + (int64_t)wg * CIN * COUT;
  if constexpr (C::SLICES > 1) {
    __syncthreads();
    float* red = stage;                                      // (SLICES-1) x BLOCKS x 32 x 32 floats <= staging size
    static_assert((C::SLICES - 1) * C::BLOCKS * 1024 * sizeof(float) <= C::LDS_BYTES, "reduction scratch fits the stage");
    if (slice > 0) {
#pragma unroll
      for (int a = 0; a < C::BI_PW; ++a)
#pragma unroll
        for (int b = 0; b < C::BO_PW; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            red[(((slice - 1) * C::BLOCKS + (bi0 + a) * C::NBO + bo0 + b) * 16 + r) * 64 + lane] = acc[a][b][r];
    }
    __syncthreads();
    if (slice == 0) {
#pragma unroll
      for (int a = 0; a < C::BI_PW; ++a)
#pragma unroll
        for (int b = 0; b < C::BO_PW; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[a][b][r];
#pragma unroll
            for (int sl = 1; sl < C::SLICES; ++sl)
              v += red[(((sl - 1) * C::BLOCKS + (bi0 + a) * C::NBO + bo0 + b) * 16 + r) * 64 + lane];
            acc[a][b][r] = v;
          }
    }
  }
  if (dbg && threadIdx.x == 0) {          // measurement runs: lifetime and placement of this workgroup
    unsigned long long* d = dbg + (size_t)wg * 4;
    d[0] = t_start;
    d[1] = wall_clock64();
    d[2] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
    d[3] = (unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) | ((unsigned long long)(nblocks - blk_lo) << 32);
  }
  if (slice == 0) {
    // C/D map of the 32x32 forms: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int a = 0; a < C::BI_PW; ++a)
#pragma unroll
      for (int b = 0; b < C::BO_PW; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ci = (bi0 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          dst[ci * COUT + (bo0 + b) * 32 + l31] = acc[a][b][r];
        }
  }
}

// ---------------------------------------------------------------------------------------------------
// wgrad v3 (tiles of 1, 2 or 4 blocks of 32x32): the four waves of a workgroup are INDEPENDENT workers. In v2 the waves of
// a workgroup share each staged step and meet at a barrier per step; with 4 workgroups per CU every SIMD hosts one wave of
// each and a workgroup only advances when all four of its waves have had their 1024-cycle MFMA turn on their SIMDs: the
// per-workgroup timeline showed the matrix pipe 62 % busy with 4 resident workgroups per CU against 73 % with 3 (convoys at
// the barriers). Here a wave owns the whole Cin x Cout tile (1/2/4 accumulator blocks), stages ITS OWN pairs in a private
// LDS double buffer (wave-level ordering only, no s_barrier in the loop), reads every staged value for 2 MFMAs instead of
// 1, and the four accumulator sets are added through LDS in a fixed order once at the end (one partial per workgroup).
template <int CIN, int COUT>
struct Wgrad3Cfg {
  static constexpr int NBI = CIN / 32, NBO = COUT / 32, BLOCKS = NBI * NBO;
  static constexpr int ROWF = CIN + COUT;
  // pairs per wave-step: 32 staging VGPRs at 32x32 / 32x64, 16 at 64x64 (3 waves per SIMD instead of 2: 156 -> 150 us)
  static constexpr int PSW = ROWF <= 64 ? 32 : (ROWF >= 128 ? 8 : 16);
  static constexpr int CHUNKS = PSW * ROWF / 4, CPT = CHUNKS / 64;      // float4 chunks per wave-step / per lane
  static constexpr int WAVE_FLOATS = 2 * PSW * ROWF;                    // private double buffer
  // the four private double buffers, or the end-of-kernel reduction scratch (3 accumulator sets) that aliases them
  static constexpr size_t LDS_BYTES = sizeof(float) * (4 * WAVE_FLOATS > 3 * CIN * COUT ? 4 * WAVE_FLOATS : 3 * CIN * COUT);
  static_assert(CHUNKS % 64 == 0, "whole chunks per lane");
};

// WINDOWED (default): the work of a workgroup is not one contiguous pair range of its offset but, for each of the WPX row
// windows of its XCD in turn, its 1/m share of the offset's pairs whose OUTPUT row lies in that window (bnd = per offset
// the first pair of every window, computed once per rulebook). All 27 offsets' workgroups of an XCD then walk the same few
// thousand rows at the same time, and an XCD only ever touches its own eighth of the rows: the gathered X / dY rows are
// fetched from HBM once per XCD window instead of once per (offset, XCD) — with the offset-major ranges of the first version
// every gathered row missed L2 (PMC: 798 MB of fabric reads per 64x64 launch against 112 MB algorithmic, 5.1 TB/s: the
// kernel ran at the memory side's speed, not the matrix pipe's).
constexpr int WG_WPX = 4;                       // row windows per XCD partition
constexpr int WG_NWIN = 8 * WG_WPX;             // windows over the output rows

template <int CIN, int COUT, int MODE = 0, bool WINDOWED = false>   // MODE (measurement builds): 1 = no MFMAs, 2 = no gather pipeline
__global__ __launch_bounds__(256) void sparse_conv_wgrad3_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                                 const int* __restrict__ pin, const int* __restrict__ pout,
                                                                 const int* __restrict__ pstart, const int* __restrict__ plan,
                                                                 float* __restrict__ partial /* (workgroup,CIN,COUT) */,
                                                                 int K, unsigned long long* __restrict__ dbg = nullptr,
                                                                 const int* __restrict__ bnd = nullptr) {
  using C = Wgrad3Cfg<CIN, COUT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* stage_all = reinterpret_cast<float*>(smem);
  const unsigned long long t_start = dbg ? wall_clock64() : 0ULL;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int wg, o, p0, p1, blk_lo, nblocks;
  int seg_a[WG_WPX], seg_e[WG_WPX], seg_t[WG_WPX + 1];      // WINDOWED: pair range and first step of every window
  if constexpr (WINDOWED) {
    const int x = blockIdx.x & 7, slot = blockIdx.x >> 3;  // XCD (hardware round-robin: locality only), slot on it
    const int S = plan[0];
    if (slot >= S) return;
    wg = x * S + slot;
    const int4 d = reinterpret_cast<const int4*>(plan + 8)[wg];
    if (d.x < 0) return;
    o = d.x;
    p0 = pstart[0];
    p1 = pstart[K];                                         // clamp range of the loads: the whole pair list
    seg_t[0] = 0;
#pragma unroll
    for (int w = 0; w < WG_WPX; ++w) {
      const int lo = bnd[o * (WG_NWIN + 1) + x * WG_WPX + w], hi = bnd[o * (WG_NWIN + 1) + x * WG_WPX + w + 1];
      const long long len = hi - lo;
      seg_a[w] = lo + (int)(len * d.y / d.z);
      seg_e[w] = lo + (int)(len * (d.y + 1) / d.z);
      seg_t[w + 1] = seg_t[w] + (seg_e[w] - seg_a[w] + 4 * C::PSW - 1) / (4 * C::PSW);
    }
    blk_lo = 0;
    nblocks = seg_t[WG_WPX];
  } else {
    wg = xcd_remap(blockIdx.x, gridDim.x);
    if (wg >= plan[1]) return;
    const int cb = plan[0];                                 // workgroup blocks (4 * PSW pairs) per workgroup
    o = 0;
    for (int k = 1; k < K; ++k)
      if (plan[2 + k] <= wg) o = k;
    p0 = pstart[o];
    p1 = pstart[o + 1];
    blk_lo = (wg - plan[2 + o]) * cb;
    nblocks = min((p1 - p0 + 4 * C::PSW - 1) / (4 * C::PSW), blk_lo + cb);
  }
  float* stage = stage_all + wave * C::WAVE_FLOATS;

  f32x16 acc[C::NBI][C::NBO];
#pragma unroll
  for (int a = 0; a < C::NBI; ++a)
#pragma unroll
    for (int b = 0; b < C::NBO; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // lane's float4 chunks of a wave-step: chunk c = lane + 64 j -> pair row c / (ROWF/4), X part first then dY part
  constexpr int RCH = C::ROWF / 4;
  auto chunk = [&](int j, int& pr, int& col, bool& isx) {
    const int c = lane + 64 * j;
    pr = c / RCH;
    const int w = c - pr * RCH;
    isx = w < CIN / 4;
    col = isx ? w : w - CIN / 4;
  };
  struct Idx { int row[C::CPT]; };
  struct Feat { f32x4 v[C::CPT]; };
  // wave-step b of this workgroup block-range: pairs p0 + (b * 4 + wave) * PSW ... (the four waves interleave, so they walk
  // neighbouring rows at the same time)
  // -> first pair of this wave's slice of step b, and (by reference) one past the last valid pair of that step's range
  auto pair_base_lim = [&](int b, int& lim) {
    if constexpr (WINDOWED) {
      int w = 0;
#pragma unroll
      for (int q = 1; q < WG_WPX; ++q) w += (b >= seg_t[q]) ? 1 : 0;
      int a = seg_a[0], e = seg_e[0], t0 = 0;
#pragma unroll
      for (int q = 1; q < WG_WPX; ++q)
        if (w == q) { a = seg_a[q]; e = seg_e[q]; t0 = seg_t[q]; }
      lim = (b < nblocks) ? e : 0;
      return a + ((b - t0) * 4 + wave) * C::PSW;
    } else {
      lim = (b < nblocks) ? p1 : 0;
      return p0 + (b * 4 + wave) * C::PSW;
    }
  };
  auto pair_base = [&](int b) { int l; return pair_base_lim(b, l); };
  auto load_idx = [&](Idx& x, int b) {
#pragma unroll
    for (int j = 0; j < C::CPT; ++j) {
      int pr, col; bool isx;
      chunk(j, pr, col, isx);
      int p = pair_base(b) + pr;
      p = p < p1 ? p : p1 - 1;
      p = p < p0 ? p0 : p;
      x.row[j] = isx ? pin[p] : pout[p];
    }
  };
  auto load_feat = [&](Feat& f, const Idx& x) {
#pragma unroll
    for (int j = 0; j < C::CPT; ++j) {
      int pr, col; bool isx;
      chunk(j, pr, col, isx);
      const float* src = isx ? X + (int64_t)x.row[j] * CIN : dY + (int64_t)x.row[j] * COUT;
      f.v[j] = *reinterpret_cast<const f32x4*>(src + col * 4);
    }
  };
  auto store_feat = [&](const Feat& f, int b, float* buf) {
#pragma unroll
    for (int j = 0; j < C::CPT; ++j) {
      int pr, col; bool isx;
      chunk(j, pr, col, isx);
      int lim;
      const int pb = pair_base_lim(b, lim);
      const bool ok = pb + pr < lim;
      f32x4 v = f.v[j];
      if (!ok) v = (f32x4){0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(buf + pr * C::ROWF + (isx ? 0 : CIN) + col * 4) = v;
    }
  };
  const int l31 = lane & 31, kh = lane >> 5;
  // One step = KS k-steps of NBI x NBO MFMAs (64 cycles each) on buffer `cur`, with everything else of the pipeline
  // INTERLEAVED between the MFMA groups so that it issues in their shadow (a lone wave that did its loads, address
  // arithmetic and LDS writes between two MFMA bursts kept the pipe 49 % busy): after the MFMAs of k-step kk the lane
  //   writes chunk j of step s+1 (in registers since step s-1) to the other buffer,
  //   re-uses the register for the gather of chunk j of step s+2 (row index loaded during step s-1),
  //   loads the row index of chunk j of step s+3.
  // Operands of k-step kk+1 are read from LDS before the MFMAs of kk (the scheduler otherwise sinks the reads).
  constexpr int KS = C::PSW / 2;
  constexpr int STEPF = C::PSW * C::ROWF;
  auto load_idx1 = [&](Idx& x, int b, int j) {
    int pr, col; bool isx;
    chunk(j, pr, col, isx);
    int p = pair_base(b) + pr;
    p = p < p1 ? p : p1 - 1;
    p = p < p0 ? p0 : p;
    x.row[j] = isx ? pin[p] : pout[p];
  };
  auto load_feat1 = [&](Feat& f, const Idx& x, int j) {
    int pr, col; bool isx;
    chunk(j, pr, col, isx);
    const float* src = isx ? X + (int64_t)x.row[j] * CIN : dY + (int64_t)x.row[j] * COUT;
    f.v[j] = *reinterpret_cast<const f32x4*>(src + col * 4);
  };
  auto store_feat1 = [&](const Feat& f, int b, float* buf, int j) {
    int pr, col; bool isx;
    chunk(j, pr, col, isx);
    int lim;
    const int pb = pair_base_lim(b, lim);
    const bool ok = pb + pr < lim;
    f32x4 v = f.v[j];
    if (!ok) v = (f32x4){0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(buf + pr * C::ROWF + (isx ? 0 : CIN) + col * 4) = v;
  };
  auto step = [&](int b, const float* buf, float* other, Feat& f, Idx& x) {
    float av[2][C::NBI], bv[2][C::NBO];
    auto fetch = [&](int kk, int s) {
      const float* row = buf + (2 * kk + kh) * C::ROWF;
#pragma unroll
      for (int a = 0; a < C::NBI; ++a) av[s][a] = row[a * 32 + l31];
#pragma unroll
      for (int bb = 0; bb < C::NBO; ++bb) bv[s][bb] = row[CIN + bb * 32 + l31];
    };
    fetch(0, 0);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      if (kk + 1 < KS) fetch(kk + 1, (kk + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < C::NBI; ++a)
#pragma unroll
        for (int bb = 0; bb < C::NBO; ++bb)
          if constexpr (MODE == 1) acc[a][bb][0] += av[kk & 1][a] * bv[kk & 1][bb];
          else acc[a][bb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk & 1][a], bv[kk & 1][bb], acc[a][bb], 0, 0, 0);
      // chunks kk*CPT/KS .. (kk+1)*CPT/KS of the pipeline, in the shadow of the MFMAs just issued
      if constexpr (MODE != 2) {
#pragma unroll
        for (int jj = (kk * C::CPT) / KS; jj < ((kk + 1) * C::CPT) / KS; ++jj) {
          store_feat1(f, b + 1, other, jj);
          load_feat1(f, x, jj);
          load_idx1(x, b + 3, jj);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // the LDS double buffer is private to the wave: its own ds_write / ds_read are executed in order, what is needed is that the
  // compiler keeps them in program order (the lanes exchange data) -> a wave-scope fence, no s_barrier
  auto wave_fence = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  Idx ix;
  Feat ft;
  int blk = blk_lo;
  if (blk < nblocks) {                       // prologue: step blk -> LDS, step blk+1 -> registers, indices of step blk+2
    load_idx(ix, blk);
    load_feat(ft, ix);
    store_feat(ft, blk, stage);
    load_idx(ix, blk + 1);
    load_feat(ft, ix);
    load_idx(ix, blk + 2);
    wave_fence();
  }
  int cur = 0;
  const unsigned long long t_loop = dbg ? __builtin_amdgcn_s_memtime() : 0ULL;
  while (blk < nblocks) {
    step(blk, stage + cur * STEPF, stage + (cur ^ 1) * STEPF, ft, ix);
    wave_fence();
    cur ^= 1;
    ++blk;
  }
  const unsigned long long loop_cycles = dbg ? __builtin_amdgcn_s_memtime() - t_loop : 0ULL;   // shader cycles of wave 0's loop

  // add the four waves' accumulators through LDS in a fixed order (scratch aliases the staging buffers: barrier first)
  __syncthreads();
  float* red = stage_all;                                    // 3 x (BLOCKS x 16 x 64) floats
  if (wave > 0) {
#pragma unroll
    for (int a = 0; a < C::NBI; ++a)
#pragma unroll
      for (int b = 0; b < C::NBO; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(((wave - 1) * C::BLOCKS + a * C::NBO + b) * 16 + r) * 64 + lane] = acc[a][b][r];
  }
  __syncthreads();
  if (dbg && threadIdx.x == 0) {
    unsigned long long* d = dbg + (size_t)wg * 4;
    d[0] = t_start;
    d[1] = wall_clock64();
    d[2] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) | (loop_cycles << 32);
    d[3] = (unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) | ((unsigned long long)(nblocks - blk_lo) << 32);
  }
  if (wave == 0) {
    float* dst = partial + (int64_t)wg * CIN * COUT;
#pragma unroll
    for (int a = 0; a < C::NBI; ++a)
#pragma unroll
      for (int b = 0; b < C::NBO; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[a][b][r];
#pragma unroll
          for (int w = 1; w < 4; ++w) v += red[(((w - 1) * C::BLOCKS + a * C::NBO + b) * 16 + r) * 64 + lane];
          const int ci = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          dst[ci * COUT + b * 32 + l31] = v;
        }
  }
}

// bnd[o][w] = first pair of offset o whose output row is >= w * ceil(n_out / WG_NWIN)  (w = 0..WG_NWIN), by binary search in
// the offset's ascending out rows. Once per rulebook.
__global__ __launch_bounds__(256) void wgrad_window_bounds_kernel(const int* __restrict__ pout, const int* __restrict__ pstart,
                                                                  int K, int n_out, int* __restrict__ bnd) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= K * (WG_NWIN + 1)) return;
  const int o = t / (WG_NWIN + 1), w = t - o * (WG_NWIN + 1);
  const int rows = (n_out + WG_NWIN - 1) / WG_NWIN;
  const long long r = (long long)w * rows;
  int lo = pstart[o], hi = pstart[o + 1];
  if (w == WG_NWIN) { bnd[t] = hi; return; }
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (pout[mid] < r) lo = mid + 1; else hi = mid;
  }
  bnd[t] = lo;
}

// windowed plan (one workgroup of 8 waves: wave x = XCD partition x, lane = kernel offset): S workgroup slots per XCD are
// dealt to the offsets in proportion to their pairs inside the partition (>= 1 for an offset with any pair), so that every
// workgroup of the XCD has about the same number of pairs per window. Layout of `plan` (ints):
//   [0] S   [8 + 4 q ..] descriptor of workgroup q = x S + slot: {offset or -1, index i, m workgroups share the offset, -}
//   [LIST0 + o] start of offset o's partial list (o = 0..K)   [LIST0 + K + 1 + k] workgroup q of the k-th partial
__device__ __forceinline__ int wgrad_plan_list0(int S) { return 8 + 4 * 8 * S; }

__global__ __launch_bounds__(512) void wgrad_window_plan_kernel(const int* __restrict__ bnd, int K, int S,
                                                                int* __restrict__ plan) {
  __shared__ int m_sh[8][32], first_sh[8][32];
  const int x = threadIdx.x >> 6, o = threadIdx.x & 63;
  int P = 0;
  if (o < K) P = bnd[o * (WG_NWIN + 1) + (x + 1) * WG_WPX] - bnd[o * (WG_NWIN + 1) + x * WG_WPX];
  long long T = P;
  for (int d = 32; d > 0; d >>= 1) T += __shfl_xor(T, d, 64);
  int m = (P > 0 && T > 0) ? (int)((long long)S * P / T) : 0;
  if (P > 0 && m < 1) m = 1;
  int sum = m;
  for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
  // too many: take from the offset with the most workgroups; too few: give to the most loaded (pairs per workgroup)
  for (int it = 0; it < 4 * 64 && sum != S && T > 0; ++it) {
    float key = sum > S ? (m > 1 ? (float)m : -1.f) : (m > 0 ? (float)P / (float)m : -1.f);
    float best = key;
    int who = o;
    for (int d = 32; d > 0; d >>= 1) {
      const float ob = __shfl_xor(best, d, 64);
      const int ow = __shfl_xor(who, d, 64);
      if (ob > best || (ob == best && ow < who)) { best = ob; who = ow; }
    }
    if (best < 0.f) break;
    if (o == who) m += sum > S ? -1 : 1;
    sum += sum > S ? -1 : 1;
  }
  int incl = m;
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(incl, d, 64);
    if (o >= d) incl += v;
  }
  const int first = incl - m;
  if (o < 32) { m_sh[x][o] = (o < K) ? m : 0; first_sh[x][o] = first; }
  if (threadIdx.x == 0) plan[0] = S;
  int4* desc = reinterpret_cast<int4*>(plan + 8);
  // slots beyond the dealt ones stay empty
  for (int q = o; q < S; q += 64) desc[x * S + q] = make_int4(-1, 0, 1, 0);
  __syncthreads();
  for (int j = 0; j < m; ++j)
    if (first + j < S) desc[x * S + first + j] = make_int4(o, j, m, 0);
  // per-offset partial lists (fixed order: XCD partition, then index)
  const int L0 = wgrad_plan_list0(S);
  if (threadIdx.x <= K) {
    const int oo = threadIdx.x;
    int start = 0;
    for (int k = 0; k < oo; ++k)
      for (int xx = 0; xx < 8; ++xx) start += m_sh[xx][k];
    plan[L0 + oo] = start;
    if (oo < K) {
      int k = start;
      for (int xx = 0; xx < 8; ++xx)
        for (int j = 0; j < m_sh[xx][oo]; ++j)
          if (first_sh[xx][oo] + j < S) plan[L0 + K + 1 + k++] = xx * S + first_sh[xx][oo] + j;
    }
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_windowed_kernel(const float* __restrict__ partial,
                                                                    const int* __restrict__ plan, float* __restrict__ dW,
                                                                    int K, int per /* CIN*COUT */) {
  __shared__ float sh[8][32];
  const int chunks = (per + 31) >> 5;
  const int o = blockIdx.x / chunks, e = (blockIdx.x - o * chunks) * 32 + (threadIdx.x & 31);
  const int part = threadIdx.x >> 5;
  const int L0 = wgrad_plan_list0(plan[0]);
  const int w0 = plan[L0 + o], w1 = plan[L0 + o + 1];
  const int* list = plan + L0 + K + 1;
  float s = 0.f;
  if (e < per)
    for (int w = w0 + part; w < w1; w += 8) s += partial[(int64_t)list[w] * per + e];
  sh[part][threadIdx.x & 31] = s;
  __syncthreads();
  if (part == 0 && e < per) {
    const int l = threadIdx.x & 31;
    dW[(int64_t)o * per + e] = ((sh[0][l] + sh[1][l]) + (sh[2][l] + sh[3][l])) + ((sh[4][l] + sh[5][l]) + (sh[6][l] + sh[7][l]));
  }
}

// dW[o] = sum of the partials of offset o's workgroups. A 256-thread workgroup owns 32 consecutive elements of one offset;
// 8 thread groups stride over the partials, then a fixed-shape LDS tree adds the 8 sums (same association every run).
// The centre offset of a SubM conv has hundreds of partials: a serial per-element loop took longer than the wgrad itself
// at C = 16.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, const int* __restrict__ plan,
                                                           float* __restrict__ dW, int K, int per /* CIN*COUT */) {
  __shared__ float sh[8][32];
  const int chunks = (per + 31) >> 5;
  const int o = blockIdx.x / chunks, e = (blockIdx.x - o * chunks) * 32 + (threadIdx.x & 31);
  const int part = threadIdx.x >> 5;
  const int w0 = plan[2 + o], w1 = plan[3 + o];
  float s = 0.f;
  if (e < per)
    for (int w = w0 + part; w < w1; w += 8) s += partial[(int64_t)w * per + e];
  sh[part][threadIdx.x & 31] = s;
  __syncthreads();
  if (part == 0 && e < per) {
    const int l = threadIdx.x & 31;
    dW[(int64_t)o * per + e] = ((sh[0][l] + sh[1][l]) + (sh[2][l] + sh[3][l])) + ((sh[4][l] + sh[5][l]) + (sh[6][l] + sh[7][l]));
  }
}

CRB_KNOB g_subt_override = 0;     // 0 = heuristic; 1/2/4 force (A/B measurements)
CRB_KNOB g_fwd_lowchannel = 1;    // sparse_conv_fwd_lc_kernel: 1 = for C_in = 4 (product); measurement builds: 0 / 2 / >= 16, see launch_fwd_compact
#ifdef CRB_MEASURE
static int g_fwd_rowc = 0;        // 1 = row-contiguous gathers + in-quad transpose at CIN = 64 (compact-table kernel)
#endif

template <int CIN, int COUT, int SUBT>
int launch_fwd_subt(const float* X, const float* W, const int* nbr, const int* perm, float* Y, int64_t n_out, int K,
                    hipStream_t st) {
  constexpr int KSTEPS = (CIN + 3) / 4;
  const int ntiles = crb_cdiv(n_out, 64 * SUBT);
  const int grid = ((ntiles + 7) / 8) * 8;
  size_t lds = 2 * sizeof(float) * KSTEPS * 4 * (((COUT + 15) / 16) * 16) + sizeof(int) * 64 * SUBT * K;
  hipLaunchKernelGGL((sparse_conv_fwd_kernel<CIN, COUT, SUBT>), dim3(grid), dim3(256), lds, st, X, W, nbr, perm, Y,
                     (int)n_out, K, ntiles);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

template <int CIN, int COUT>
int launch_fwd(const float* X, const float* W, const int* nbr, const int* perm, float* Y, int64_t n_out, int K,
               hipStream_t st) {
  // rows per workgroup = 64*subt. Measured on the SECOND bs=16 geometry (tools/bench_sparse_conv.py): subt 1/2/4 =
  // 221/258/343 us at C=64, 96/115/160 us at C=32 — the kernel is latency-, not W-traffic-bound, so more, smaller
  // workgroups win. subt > 1 stays available for measurements only.
  int subt = (g_subt_override == 2 || g_subt_override == 4) ? g_subt_override : 1;
  if constexpr (CIN % 16 == 0 && COUT % 16 == 0 && CIN <= 64) {
    if (g_subt_override == 0 || g_subt_override == 8 || g_subt_override == 9 || g_subt_override == 16 ||
        g_subt_override == 32) {   // v2
      const int ntiles = crb_cdiv(n_out, 64);
      const int grid = ((ntiles + 7) / 8) * 8;
      size_t lds = 2 * sizeof(float) * CIN * (((COUT + 15) / 16) * 16) + sizeof(int) * 64 * K;
      if constexpr (CIN == 64 && COUT == 64) {
        if (g_subt_override == 16) {                 // measurement: tiles in blockIdx order (no XCD-contiguous remap)
          hipLaunchKernelGGL((sparse_conv_fwd2_kernel<CIN, COUT, false, false>), dim3(grid), dim3(256), lds, st, X, W, nbr,
                             perm, Y, (int)n_out, K, ntiles);
          CRB_CHECK_LAUNCH();
          return CRB_OK;
        }
        if (g_subt_override == 32) {                 // measurement: per-region cycle accounting (crb_sparse_conv_timing)
          hipLaunchKernelGGL((sparse_conv_fwd2_kernel<CIN, COUT, false, true, true>), dim3(grid), dim3(256), lds, st, X, W,
                             nbr, perm, Y, (int)n_out, K, ntiles);
          CRB_CHECK_LAUNCH();
          return CRB_OK;
        }
        if (g_subt_override == 9) {                  // staging-only measurement variant (wrong results by design)
          hipLaunchKernelGGL((sparse_conv_fwd2_kernel<CIN, COUT, true>), dim3(grid), dim3(256), lds, st, X, W, nbr, perm,
                             Y, (int)n_out, K, ntiles);
          CRB_CHECK_LAUNCH();
          return CRB_OK;
        }
      }
      hipLaunchKernelGGL((sparse_conv_fwd2_kernel<CIN, COUT>), dim3(grid), dim3(256), lds, st, X, W, nbr, perm, Y,
                         (int)n_out, K, ntiles);
      CRB_CHECK_LAUNCH();
      return CRB_OK;
    }
  }
  if constexpr (CIN >= 16 && COUT <= 64) {
    if (subt == 4) return launch_fwd_subt<CIN, COUT, 4>(X, W, nbr, perm, Y, n_out, K, st);
    if (subt == 2) return launch_fwd_subt<CIN, COUT, 2>(X, W, nbr, perm, Y, n_out, K, st);
  }
  return launch_fwd_subt<CIN, COUT, 1>(X, W, nbr, perm, Y, n_out, K, st);
}

template <int CIN, int COUT>
int launch_fwd_compact(const float* X, const float* W, const unsigned* cmask, const int* cbase, const int* packed,
                       const int* perm, const int* tile_order, float* Y, int64_t n_out, int K, hipStream_t st,
                       ConvEpilogue ep = ConvEpilogue{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0}) {
  if constexpr ((CIN == 4 || CIN == 16) && COUT == 16) {
    // g_fwd_lowchannel: 1 (product) = the low-channel kernel for C_in = 4 only (20.1 us against 25.4 for the v1 kernel on the
    // (N,K) table; at 16 -> 16 it measured 28.5 us against 17.2 for the phase kernel: profiles/r04_lowchannel_floor.txt);
    // measurement builds: 0 = never, 2 = both shapes, >= 16 = both shapes with the weights resident in LDS, that many workgroups
    const bool use_lc = g_fwd_lowchannel >= 2 || (g_fwd_lowchannel == 1 && CIN == 4);
    if (use_lc) {
      const int ntiles = crb_cdiv(n_out, 16);
      if (g_fwd_lowchannel < 16) {                           // weights from L1 / L2, one tile per wave, 4 waves per workgroup
        const int grid = crb_cdiv(ntiles, 4);
        hipLaunchKernelGGL((sparse_conv_fwd_lc_kernel<CIN, 8, false>), dim3(grid), dim3(256), sizeof(int) * 4 * (16 * 32 + 16), st, X,
                           W, cmask, cbase, packed, perm, Y, (int)n_out, K, ntiles, ep);
      } else {
#ifdef CRB_MEASURE
        int grid = crb_cdiv(ntiles, 16);                     // >= one tile per wave
        if (grid > g_fwd_lowchannel) grid = g_fwd_lowchannel;
        const size_t lds = sizeof(float) * ((K * CIN * COUT + 3) & ~3) + sizeof(int) * 16 * (16 * 32 + 16);
        CRB_HIP(hipFuncSetAttribute((const void*)sparse_conv_fwd_lc_kernel<CIN, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        hipLaunchKernelGGL((sparse_conv_fwd_lc_kernel<CIN, 8, true>), dim3(grid), dim3(1024), lds, st, X, W, cmask, cbase, packed, perm,
                           Y, (int)n_out, K, ntiles, ep);
#endif
      }
      CRB_CHECK_LAUNCH();
      return CRB_OK;
    }
  }
  if constexpr (CIN % 16 == 0 && COUT % 16 == 0 && CIN <= 64) {
    const int ntiles = crb_cdiv(n_out, 64);
    const int grid = ((ntiles + 7) / 8) * 8;
    size_t lds = 2 * sizeof(float) * CIN * (((COUT + 15) / 16) * 16) + sizeof(int) * 64 * K + 16;
#ifdef CRB_MEASURE
    // measured variant, not in the product library (tools/ab_knob.py crb_sparse_conv_set_rowc: 162.5 vs 147.2 us at L3, 110.6
    // vs 100.0 us at L4 — the cache-line look-ups it saves are not on this kernel's critical path, its ~100 extra DPP / select
    // / address operations per phase and their hazards are)
    if constexpr (CIN == 64) {
      if (g_fwd_rowc) {
        hipLaunchKernelGGL((sparse_conv_fwd2_kernel<CIN, COUT, false, true, false, true, true>), dim3(grid), dim3(256), lds, st,
                           X, W, packed, perm, Y, (int)n_out, K, ntiles, cmask, cbase, ep, tile_order);
        CRB_CHECK_LAUNCH();
        return CRB_OK;
      }
    }
#endif
    hipLaunchKernelGGL((sparse_conv_fwd2_kernel<CIN, COUT, false, true, false, true>), dim3(grid), dim3(256), lds, st, X, W,
                       packed, perm, Y, (int)n_out, K, ntiles, cmask, cbase, ep, tile_order);
    CRB_CHECK_LAUNCH();
    return CRB_OK;
  }
  return CRB_ERR_UNSUPPORTED;
}


// windowed v3 launch: S = resident workgroup slots per XCD; partial (8 S, CIN, COUT); plan ints >= wgrad_windowed_plan_ints(S)
static inline int64_t wgrad_windowed_plan_ints(int S, int K) { return 8 + 4 * 8 * (int64_t)S + (K + 1) + 8 * (int64_t)S + 64; }

template <int CIN, int COUT>
int launch_wgrad_windowed(const float* X, const float* dY, const int* pin, const int* pout, const int* pstart,
                          const int* bnd, float* dW, float* partial, int* plan, int K, int S, hipStream_t st) {
  if constexpr (CIN % 32 == 0 && COUT % 32 == 0 && (CIN / 32) * (COUT / 32) <= 4) {
    using C3 = Wgrad3Cfg<CIN, COUT>;
    hipLaunchKernelGGL(wgrad_window_plan_kernel, dim3(1), dim3(512), 0, st, bnd, K, S, plan);
    if constexpr (CIN == 64 && COUT == 64) {
      if (g_wgrad_mode == 1)
        hipLaunchKernelGGL((sparse_conv_wgrad3_kernel<CIN, COUT, 1, true>), dim3(8 * S), dim3(256), C3::LDS_BYTES, st, X, dY,
                           pin, pout, pstart, plan, partial, K, g_wgrad_dbg, bnd);
      if (g_wgrad_mode == 2)
        hipLaunchKernelGGL((sparse_conv_wgrad3_kernel<CIN, COUT, 2, true>), dim3(8 * S), dim3(256), C3::LDS_BYTES, st, X, dY,
                           pin, pout, pstart, plan, partial, K, g_wgrad_dbg, bnd);
    }
    if (g_wgrad_mode == 0 || !(CIN == 64 && COUT == 64))
      hipLaunchKernelGGL((sparse_conv_wgrad3_kernel<CIN, COUT, 0, true>), dim3(8 * S), dim3(256), C3::LDS_BYTES, st, X, dY,
                         pin, pout, pstart, plan, partial, K, g_wgrad_dbg, bnd);
    hipLaunchKernelGGL(wgrad_reduce_windowed_kernel, dim3(K * ((CIN * COUT + 31) / 32)), dim3(256), 0, st, partial, plan, dW,
                       K, CIN * COUT);
    CRB_CHECK_LAUNCH();
    return CRB_OK;
  }
  return CRB_ERR_UNSUPPORTED;
}

template <int CIN, int COUT>
int launch_wgrad(const float* X, const float* dY, const int* pin, const int* pout, const int* pstart, float* dW,
                 float* partial, int* plan, int K, int S, hipStream_t st) {
  if constexpr (CIN % 32 == 0 && COUT % 32 == 0 && (CIN / 32) * (COUT / 32) <= 4) {
    if (g_wgrad_v1 == 0) {
      using C3 = Wgrad3Cfg<CIN, COUT>;
      const int maxwg = K * S;
      constexpr int shift = C3::PSW == 32 ? 7 : (C3::PSW == 16 ? 6 : 5);   // workgroup block = 4 waves x PSW pairs
      hipLaunchKernelGGL(wgrad_plan_kernel, dim3(1), dim3(64), 0, st, pstart, K, maxwg - K, plan, shift);
      if constexpr (CIN == 64 && COUT == 64) {
        if (g_wgrad_mode == 1)
          hipLaunchKernelGGL((sparse_conv_wgrad3_kernel<CIN, COUT, 1>), dim3(((maxwg + 7) / 8) * 8), dim3(256), C3::LDS_BYTES,
                             st, X, dY, pin, pout, pstart, plan, partial, K, g_wgrad_dbg);
        if (g_wgrad_mode == 2)
          hipLaunchKernelGGL((sparse_conv_wgrad3_kernel<CIN, COUT, 2>), dim3(((maxwg + 7) / 8) * 8), dim3(256), C3::LDS_BYTES,
                             st, X, dY, pin, pout, pstart, plan, partial, K, g_wgrad_dbg);
      }
      if (g_wgrad_mode == 0 || !(CIN == 64 && COUT == 64))
        hipLaunchKernelGGL((sparse_conv_wgrad3_kernel<CIN, COUT>), dim3(((maxwg + 7) / 8) * 8), dim3(256), C3::LDS_BYTES, st,
                           X, dY, pin, pout, pstart, plan, partial, K, g_wgrad_dbg);
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(K * ((CIN * COUT + 31) / 32)), dim3(256), 0, st, partial, plan, dW, K,
                         CIN * COUT);
      CRB_CHECK_LAUNCH();
      return CRB_OK;
    }
  }
  if constexpr (CIN % 32 == 0 && COUT % 32 == 0) {
    if (g_wgrad_v1 != 1) {
      using C2 = Wgrad2Cfg<CIN, COUT>;
      const int maxwg = K * S;
      hipLaunchKernelGGL(wgrad_plan_kernel, dim3(1), dim3(64), 0, st, pstart, K, maxwg - K, plan, C2::PS == 64 ? 6 : 5);
      hipLaunchKernelGGL((sparse_conv_wgrad2_kernel<CIN, COUT>), dim3(((maxwg + 7) / 8) * 8), dim3(256), C2::LDS_BYTES, st,
                         X, dY, pin, pout, pstart, plan, partial, K, g_wgrad_dbg);
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(K * ((CIN * COUT + 31) / 32)), dim3(256), 0, st, partial, plan, dW, K,
                         CIN * COUT);
      CRB_CHECK_LAUNCH();
      return CRB_OK;
    }
  }
  using C = WgradCfg<CIN, COUT>;
  size_t lds = C::SLICES > 1 ? sizeof(float) * (C::SLICES - 1) * C::NCI * 16 * C::NB * 16 : 0;
  const int maxwg = K * S;                                   // partial slots; the plan aims at maxwg - K workgroups
  hipLaunchKernelGGL(wgrad_plan_kernel, dim3(1), dim3(64), 0, st, pstart, K, maxwg - K, plan);
  hipLaunchKernelGGL((sparse_conv_wgrad_kernel<CIN, COUT>), dim3(((maxwg + 7) / 8) * 8), dim3(256), lds, st, X, dY, pin,
                     pout, pstart, plan, partial, K);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(K * ((CIN * COUT + 31) / 32)), dim3(256), 0, st, partial, plan, dW, K,
                     CIN * COUT);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

}  // namespace

#define CRB_CONV_SHAPES(X_) \
  X_(4, 16) X_(5, 16) X_(16, 4) X_(16, 5) X_(4, 64) X_(64, 4) X_(16, 16) X_(16, 32) X_(32, 16) X_(32, 32) X_(32, 64) X_(64, 32) X_(64, 64) X_(64, 128) \
  X_(128, 64) X_(128, 128)

extern "C" int crb_sparse_conv_supported(int cin, int cout) {
#define X_(a, b) if (cin == a && cout == b) return 1;
  CRB_CONV_SHAPES(X_)
#undef X_
  return 0;
}

#ifdef CRB_MEASURE
extern "C" int crb_sparse_conv_set_rowc(int on) {
  g_fwd_rowc = on ? 1 : 0;
  return CRB_OK;
}
extern "C" int crb_sparse_conv_set_lowchannel(int v) {               // 1 = product default; 0 / 2 / >= 16: launch_fwd_compact
  g_fwd_lowchannel = v < 0 ? 1 : v;
  return CRB_OK;
}

extern "C" int crb_sparse_conv_set_subtiles(int subt) {
  // 0 = default (v2 where the shape allows, else v1); 1,2,4 = v1 with that many row tiles per wave; 8 = v2 (A/B runs)
  g_subt_override = (subt == 1 || subt == 2 || subt == 4 || subt == 8 || subt == 9 || subt == 16 || subt == 32) ? subt : 0;
  return CRB_OK;
}

// measurement builds only: read (and clear) the cycle counters accumulated by set_subtiles(32) launches of the 64x64 kernel
extern "C" int crb_sparse_conv_timing(uint64_t* out16_host) {
  unsigned long long h[16];
  CRB_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fwd2_timing), sizeof(h)));
  for (int k = 0; k < 16; ++k) out16_host[k] = h[k];
  unsigned long long z[16] = {0};
  CRB_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_fwd2_timing), z, sizeof(z)));
  return CRB_OK;
}
#endif

extern "C" int crb_nbr_masks(const int32_t* nbr, int64_t n, int K, int32_t* mask, void* stream) {
  if (n < 0 || K <= 0 || K > 32) return CRB_ERR_ARG;
  if (n == 0) return CRB_OK;
  hipLaunchKernelGGL(nbr_mask_kernel, dim3(crb_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, nbr, (int)n, K, mask);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_nbr_permute(const int32_t* nbr, const int32_t* perm, int64_t n, int K, int32_t* nbr_sorted,
                               void* stream) {
  if (n < 0 || K <= 0) return CRB_ERR_ARG;
  if (n == 0) return CRB_OK;
  hipLaunchKernelGGL(nbr_permute_kernel, dim3(crb_cdiv(n * K, 256)), dim3(256), 0, (hipStream_t)stream, nbr, perm,
                     (int)n, K, nbr_sorted);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_sparse_conv_forward(const float* X, const float* W, const int32_t* nbr, const int32_t* perm, float* Y,
                                       int64_t n_out, int K, int cin, int cout, void* stream) {
  if (n_out < 0 || K <= 0 || K > 32) return CRB_ERR_ARG;
  if (n_out == 0) return CRB_OK;
  hipStream_t st = (hipStream_t)stream;
#define X_(a, b) if (cin == a && cout == b) return launch_fwd<a, b>(X, W, nbr, perm, Y, n_out, K, st);
  CRB_CONV_SHAPES(X_)
#undef X_
  return CRB_ERR_UNSUPPORTED;
}

#ifdef CRB_MEASURE
extern "C" int crb_sparse_conv_set_wgrad_v1(int on) { g_wgrad_v1 = (on == 1 || on == 2) ? on : 0; return CRB_OK; }
extern "C" int crb_sparse_conv_set_wgrad_mode(int mode) { g_wgrad_mode = (mode == 1 || mode == 2) ? mode : 0; return CRB_OK; }
extern "C" int crb_sparse_conv_set_wgrad_debug(void* dev_buf_u64x4_per_wg) {
  g_wgrad_dbg = (unsigned long long*)dev_buf_u64x4_per_wg;
  return CRB_OK;
}
#endif

// measurement helper: resident workgroups per CU of the wgrad kernel instance the dispatcher would pick (-1: no instance)
template <int A, int B>
static int wgrad_occ_of() {
  int n = -1;
  if constexpr (A % 32 == 0 && B % 32 == 0 && (A / 32) * (B / 32) <= 4) {
    if (g_wgrad_v1 == 0) {
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sparse_conv_wgrad3_kernel<A, B>, 256,
                                                       Wgrad3Cfg<A, B>::LDS_BYTES) != hipSuccess) n = -1;
      return n;
    }
  }
  if constexpr (A % 32 == 0 && B % 32 == 0) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sparse_conv_wgrad2_kernel<A, B>, 256,
                                                     Wgrad2Cfg<A, B>::LDS_BYTES) != hipSuccess) n = -1;
  }
  return n;
}
extern "C" int crb_sparse_conv_wgrad_occupancy(int cin, int cout) {
#define X_(a, b) if (cin == a && cout == b) return wgrad_occ_of<a, b>();
  X_(32, 32) X_(32, 64) X_(64, 32) X_(64, 64) X_(64, 128) X_(128, 64) X_(128, 128)
#undef X_
  return -1;
}

extern "C" int crb_sparse_conv_compact_supported(int cin, int cout) {
  if (cin == 4 && cout == 16) return 1;                     // low-channel instance (sparse_conv_fwd_lc_kernel)
  return (cin % 16 == 0 && cout % 16 == 0 && cin <= 64 && crb_sparse_conv_supported(cin, cout)) ? 1 : 0;
}

extern "C" int64_t crb_nbr_compact_workspace_bytes(int64_t n) {
  return (int64_t)sizeof(int) * (crb_scan_num_tiles(n) + 8) + 256;
}

extern "C" int crb_nbr_compact(const int32_t* nbr, const int32_t* perm, int64_t n, int K, uint32_t* cmask, int32_t* cbase,
                               int32_t* packed, void* workspace, int64_t workspace_bytes, void* stream) {
  if (n < 0 || K <= 0 || K > 32) return CRB_ERR_ARG;
  if (workspace_bytes < crb_nbr_compact_workspace_bytes(n) || !workspace) return CRB_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) { CRB_HIP(hipMemsetAsync(cbase, 0, sizeof(int), st)); return CRB_OK; }
  const int blocks = crb_cdiv(n * 32, 256);
  hipLaunchKernelGGL(nbr_compact_mask_kernel, dim3(blocks), dim3(256), 0, st, nbr, perm, (int)n, K, cmask);
  int* tiles = (int*)workspace;
  const unsigned* cm = cmask;
  int* cb = cbase;
  auto f = [cm] __device__(int64_t i) { return __popc(cm[i]); };
  auto w = [cb] __device__(int64_t i, int ex, int v) { cb[i] = ex; };
  int rc = crb_device_excl_scan(f, w, n, tiles, cbase + n, st);
  if (rc != CRB_OK) return rc;
  hipLaunchKernelGGL(nbr_compact_fill_kernel, dim3(blocks), dim3(256), 0, st, nbr, perm, (int)n, K, cmask, cbase, packed);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_sparse_conv_forward_compact(const float* X, const float* W, const uint32_t* cmask, const int32_t* cbase,
                                               const int32_t* packed, const int32_t* perm, const int32_t* tile_order, float* Y,
                                               int64_t n_out, int K, int cin, int cout, void* stream) {
  if (n_out < 0 || K <= 0 || K > 32) return CRB_ERR_ARG;
  if (n_out == 0) return CRB_OK;
  hipStream_t st = (hipStream_t)stream;
#define X_(a, b) if (cin == a && cout == b) return launch_fwd_compact<a, b>(X, W, cmask, cbase, packed, perm, tile_order, Y, n_out, K, st);
  CRB_CONV_SHAPES(X_)
#undef X_
  return CRB_ERR_UNSUPPORTED;
}

extern "C" int crb_sparse_conv_forward_compact_bn(const float* X, const float* W, const uint32_t* cmask, const int32_t* cbase,
                                                  const int32_t* packed, const int32_t* perm, const int32_t* tile_order,
                                                  float* Y, int64_t n_out, int K,
                                                  int cin, int cout, const float* bias, const float* gamma, const float* beta,
                                                  const float* running_mean, const float* running_var, float eps, int relu,
                                                  void* stream) {
  if (n_out < 0 || K <= 0 || K > 32 || !gamma || !beta || !running_mean || !running_var) return CRB_ERR_ARG;
  if (n_out == 0) return CRB_OK;
  hipStream_t st = (hipStream_t)stream;
  const ConvEpilogue ep{bias, gamma, beta, running_mean, running_var, eps, relu};
#define X_(a, b) \
  if (cin == a && cout == b) return launch_fwd_compact<a, b>(X, W, cmask, cbase, packed, perm, tile_order, Y, n_out, K, st, ep);
  CRB_CONV_SHAPES(X_)
#undef X_
  return CRB_ERR_UNSUPPORTED;
}

CRB_KNOB g_wgrad_splits = 96;   // workgroups per kernel offset the plan aims at (multiple of 8: XCD mapping)
extern "C" int crb_sparse_conv_wgrad_splits(void) { return g_wgrad_splits; }
// 16x16 tiles: the partial reduction costs as much as the MFMA work, fewer and larger workgroups win (sweep on the SECOND
// bs=16 geometry: 32 / 96 / 256 workgroups per offset = 39 / 52 / 95 us at C=16, 292 / 264 / 259 us at C=64)
// v2 shapes (both multiples of 32): ONE round of workgroups, all resident from the start. A per-workgroup timeline
// (tools/wgrad_timeline.py) of the 64x64 layer with 1058 workgroups showed 1010 starting at t = 0 and 48 queueing ~100 us for
// a slot, then running alone: kernel span 183 us for a median workgroup lifetime of 118 us. The occupancy API answers 5
// workgroups per CU there (160 KB / 32 KB of LDS) but 4 run, so the launch is sized for (API - 1) x CUs slots; each extra
// workgroup is also one more Cin x Cout partial to write and reduce.
template <int A, int B>
static size_t wgrad_lds_of() {
  if constexpr (A % 32 == 0 && B % 32 == 0 && (A / 32) * (B / 32) <= 4) {
    if (g_wgrad_v1 == 0) return Wgrad3Cfg<A, B>::LDS_BYTES;
  }
  if constexpr (A % 32 == 0 && B % 32 == 0) return Wgrad2Cfg<A, B>::LDS_BYTES;
  return 0;
}
static int wgrad2_slots(int cin, int cout) {
  static int cache[3][5][5];                                // [kernel-selection knob][cin/32][cout/32]
  int& c = cache[g_wgrad_v1][cin / 32][cout / 32];
  if (c == 0) {
    int occ = crb_sparse_conv_wgrad_occupancy(cin, cout);
    size_t lds = 0;
#define X_(a, b) if (cin == a && cout == b) lds = wgrad_lds_of<a, b>();
    X_(32, 32) X_(32, 64) X_(64, 32) X_(64, 64) X_(64, 128) X_(128, 64) X_(128, 128)
#undef X_
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      cus = 256;
    // the API answers 160 KB / LDS-per-workgroup, but a workgroup set that fills the CU's LDS to the last byte is not
    // admitted (5 x 32 KB: 4 run) -> leave 1 KB
    const int fit = lds ? (int)((160 * 1024 - 1024) / lds) : occ;
    if (occ > fit) occ = fit;
    if (occ < 1) occ = 1;
    c = occ * cus;
  }
  return c;
}
static inline int wgrad_splits_for(int K, int cin, int cout) {
  if (g_wgrad_splits != 96) return g_wgrad_splits;
  if (cin * cout <= 256) return 32;
  if (cin % 32 == 0 && cout % 32 == 0 && cin <= 128 && cout <= 128 && g_wgrad_v1 != 1) {
    int s = (wgrad2_slots(cin, cout) - 8) / K;               // K*s workgroup slots, of which K are rounding slack of the plan
    return s < 4 ? 4 : (s > 96 ? 96 : s);
  }
  return 96;
}
#ifdef CRB_MEASURE
extern "C" int crb_sparse_conv_set_wgrad_splits(int s) {      // A/B measurements; 0 restores the default
  g_wgrad_splits = (s >= 8 && s <= 1024 && s % 8 == 0) ? s : 96;
  return CRB_OK;
}
#endif

extern "C" int64_t crb_sparse_conv_wgrad_workspace_bytes(int K, int cin, int cout) {
  return (int64_t)wgrad_splits_for(K, cin, cout) * K * cin * cout * 4 + 256;
}

extern "C" int crb_wgrad_num_windows(void) { return WG_NWIN; }

extern "C" int crb_wgrad_window_bounds(const int32_t* pair_out, const int32_t* pair_start, int K, int64_t n_out,
                                       int32_t* bounds /* K * (crb_wgrad_num_windows() + 1) */, void* stream) {
  if (K <= 0 || K > 32 || n_out < 0) return CRB_ERR_ARG;
  hipLaunchKernelGGL(wgrad_window_bounds_kernel, dim3(crb_cdiv(K * (WG_NWIN + 1), 256)), dim3(256), 0, (hipStream_t)stream,
                     pair_out, pair_start, K, (int)n_out, bounds);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_sparse_conv_wgrad_windowed_supported(int cin, int cout) {
  return (cin % 32 == 0 && cout % 32 == 0 && (cin / 32) * (cout / 32) <= 4 && crb_sparse_conv_supported(cin, cout) &&
          g_wgrad_v1 == 0) ? 1 : 0;
}

static inline int wgrad_windowed_slots_per_xcd(int cin, int cout) {
  int s = wgrad2_slots(cin, cout) / 8;
  return s < 1 ? 1 : (s > 256 ? 256 : s);
}

extern "C" int64_t crb_sparse_conv_wgrad_windowed_workspace_bytes(int K, int cin, int cout) {
  const int S = wgrad_windowed_slots_per_xcd(cin, cout);
  return (int64_t)8 * S * cin * cout * 4 + wgrad_windowed_plan_ints(S, K) * 4 + 512;
}

extern "C" int crb_sparse_conv_wgrad_windowed(const float* X, const float* dY, const int32_t* pair_in,
                                              const int32_t* pair_out, const int32_t* pair_start, const int32_t* bounds,
                                              float* dW, int K, int cin, int cout, void* workspace,
                                              int64_t workspace_bytes, void* stream) {
  if (K <= 0 || K > 32) return CRB_ERR_ARG;
  if (!crb_sparse_conv_wgrad_windowed_supported(cin, cout)) return CRB_ERR_UNSUPPORTED;
  if (workspace_bytes < crb_sparse_conv_wgrad_windowed_workspace_bytes(K, cin, cout) || !workspace) return CRB_ERR_WORKSPACE;
  const int S = wgrad_windowed_slots_per_xcd(cin, cout);
  if (S < K) return CRB_ERR_UNSUPPORTED;                    // every offset needs its own workgroup on every XCD
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace;
  int* plan = (int*)((char*)workspace + crb_align_up((int64_t)8 * S * cin * cout * 4, 256));
#define X_(a, b)                                                                                              \
  if (cin == a && cout == b)                                                                                  \
    return launch_wgrad_windowed<a, b>(X, dY, pair_in, pair_out, pair_start, bounds, dW, partial, plan, K, S, st);
  CRB_CONV_SHAPES(X_)
#undef X_
  return CRB_ERR_UNSUPPORTED;
}

extern "C" int crb_sparse_conv_wgrad(const float* X, const float* dY, const int32_t* pair_in, const int32_t* pair_out,
                                     const int32_t* pair_start, float* dW, int K, int cin, int cout,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
  if (K <= 0 || K > 32) return CRB_ERR_ARG;
  if (workspace_bytes < crb_sparse_conv_wgrad_workspace_bytes(K, cin, cout) || !workspace) return CRB_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int S = wgrad_splits_for(K, cin, cout);
#define X_(a, b)                                                                                              \
  if (cin == a && cout == b)                                                                                  \
    return launch_wgrad<a, b>(X, dY, pair_in, pair_out, pair_start, dW, (float*)workspace,                     \
                              (int*)((char*)workspace + (int64_t)S * K * cin * cout * 4), K, S, st);
  CRB_CONV_SHAPES(X_)
#undef X_
  return CRB_ERR_UNSUPPORTED;
}

// ---- weight layouts of all layers of a step in one launch -----------------------------------------------------------------------
// spconv keeps a layer's weight as (Cout, K, Cin) (the checkpoint layout); the gather-GEMM multiplies by W[o] (Cin, Cout) forward and
// by W[o]^T (flipped over o for submanifold layers) in the input gradient. Per layer that is a permuted copy forward and a flip + a
// transposed copy backward: three launch-bound launches of < 0.5 MB each, 31 per SECOND step. One launch writes all of them.
namespace {
constexpr int SWJ_MAX = 32;
struct SparseWJob { const float* w; float* kio; float* wd; int K, cin, cout, flip, first_block; };
struct SparseWJobs { int n; SparseWJob job[SWJ_MAX]; };

__global__ __launch_bounds__(256) void sparse_weights_multi_kernel(SparseWJobs jobs) {
  int j = 0;
  while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.job[j + 1].first_block) ++j;
  const SparseWJob& q = jobs.job[j];
  const int64_t t = (int64_t)(blockIdx.x - q.first_block) * 256 + threadIdx.x;      // index into kio: ((k, i), o)
  const int64_t total = (int64_t)q.K * q.cin * q.cout;
  if (t >= total) return;
  const int o = (int)(t % q.cout);
  const int i = (int)((t / q.cout) % q.cin);
  const int k = (int)(t / ((int64_t)q.cout * q.cin));
  const float v = q.w[((int64_t)o * q.K + k) * q.cin + i];
  q.kio[t] = v;                                                                     // W[k][i][o]
  if (q.wd) q.wd[((int64_t)(q.flip ? q.K - 1 - k : k) * q.cout + o) * q.cin + i] = v;   // Wd[k'][o][i] = W[k][i][o], k' = K-1-k if flip
}
}  // namespace

extern "C" int crb_sparse_weights_multi(int n, const float* const* w, const int32_t* K, const int32_t* cin, const int32_t* cout,
                                        const int32_t* flip, float* const* w_kio, float* const* w_dgrad, void* stream) {
  if (n < 0 || n > SWJ_MAX || (n > 0 && (!w || !K || !cin || !cout || !flip || !w_kio || !w_dgrad))) return CRB_ERR_ARG;
  if (n == 0) return CRB_OK;
  SparseWJobs jobs;
  jobs.n = n;
  int64_t blocks = 0;
  for (int j = 0; j < n; ++j) {
    if (!w[j] || !w_kio[j] || K[j] <= 0 || cin[j] <= 0 || cout[j] <= 0) return CRB_ERR_ARG;
    jobs.job[j] = SparseWJob{w[j], w_kio[j], w_dgrad[j], K[j], cin[j], cout[j], flip[j] ? 1 : 0, (int)blocks};
    blocks += crb_cdiv((int64_t)K[j] * cin[j] * cout[j], 256);
    if (blocks >= (1LL << 30)) return CRB_ERR_ARG;
  }
  hipLaunchKernelGGL(sparse_weights_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, jobs);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}
