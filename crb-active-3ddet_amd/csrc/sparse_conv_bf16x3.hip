// Opt-in split-bf16 ("bf16x3") arithmetic for the output-stationary gather-GEMM of sparse_conv.hip (row a4 of SURVEY §8).
//
// The exact-f32 MFMA (v_mfma_f32_16x16x4_f32) runs at 1/16 of the bf16 matrix rate: at C = 64 the forward kernel's MFMA roof
// (65.6 us on the SECOND bs=16 level-3 table) sits 4-5x above its HBM roof (14 us at 8 TB/s). This file trades the last
// bits of the products for that factor, as a STATED contract the caller has to ask for (crb_sparse_conv_forward_bf16x3;
// crb_sparse_conv_forward / _compact stay exact f32 and stay the default):
//
//   x = x_hi + x_lo + rx,  x_hi = bf16_rne(x), x_lo = bf16_rne(x - x_hi)  (x - x_hi is exact in f32), |rx| <= 2^-18 |x|
//   x*w ~= x_lo*w_hi + x_hi*w_lo + x_hi*w_hi     three v_mfma_f32_16x16x32_bf16 passes, products exact, f32 accumulate
//   dropped: x_lo*w_lo (<= 2^-18 |x w|) and the two residuals (<= 2^-18 |x w| each)
//   =>  |y_bf16x3 - y_exact| <= 2^-16 * sum_o sum_k |x_k| |w_k|  + the usual f32 accumulation error of either kernel.
// bf16 keeps f32's exponent range, so there is no scaling, overflow or underflow case beyond f32's own (x_lo of a value
// below 2^-118 flushes: absolute error < 2^-126, covered by the bound in practice and stated here).
//
// Same tables (mask + packed indices, rows in mask-sorted LPT order), same phase structure as sparse_conv_fwd2_kernel: a
// workgroup walks the kernel offsets present in its rows; per phase W[o] (pre-split into bf16 hi/lo vectors laid out in the
// order the lanes read them: one straight 16-byte-per-lane copy global -> LDS, conflict-free ds_read_b128) is handed over
// through a double buffer with one barrier, the rows of the NEXT offset are gathered into registers under this phase's MFMAs,
// converted to hi/lo bf16 in registers when the phase starts (VALU, idle otherwise). A wave owns TPW 16-row tiles that
// share every B fragment read from LDS (at 16 rows per wave the kernel would be LDS-read-bound: 16 KB of W per 16 rows and
// phase against 96 MFMA cycles).
#ifdef CRB_MEASURE   // the opt-in split-bf16 gather-GEMM (round 2): it buys nothing on the default path; MEASUREMENT library only since round 5
#include "crb_common.h"
#include "../../include/crb_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

CRB_KNOB g_bf16x3_tpw = 0;        // measurement knob: 0 = default, 1 / 2 = tiles per wave
CRB_KNOB g_bf16x3_mode = 0;       // measurement builds of the 64x64 one-tile kernel (see MODE)

namespace {

__device__ __forceinline__ int xcd_remap(int b, int nblocks) {
  const int per = (nblocks + 7) >> 3;      // block b runs on XCD b % 8: give every XCD a contiguous chunk of tiles
  return (b & 7) * per + (b >> 3);
}

__device__ __forceinline__ void split8(const u32x4& v0, const u32x4& v1, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = __uint_as_float(e < 4 ? v0[e & 3] : v1[e & 3]);
    const __bf16 h = (__bf16)x;
    hi[e] = h;
    lo[e] = (__bf16)(x - (float)h);
  }
}

template <int CTRL>
__device__ __forceinline__ u32x4 dpp4(const u32x4& v) {          // every component from the quad lane CTRL names
  u32x4 r;
#pragma unroll
  for (int t = 0; t < 4; ++t) r[t] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v[t], CTRL, 0xF, 0xF, false);
  return r;
}

// k index of element e of lane group g in k-group s (see w_split_pack_kernel). CIN = 64 uses the row-contiguous gather
// (each lane ends up with 16 consecutive channels of its row), the other shapes the fragment-shaped one.
template <int CIN>
__device__ __forceinline__ int k_of(int s, int g, int e) {
  return CIN == 64 ? 16 * g + 8 * s + e : 32 * s + 16 * (e >> 2) + 4 * g + (e & 3);
}

// Wp layout per offset o: [s = k-group of 32][nb = 16-column block][hl: 0 hi, 1 lo][lane = 16 g + li] -> 8 bf16 =
// W[o][k(s, g, e)][16 nb + li], e = 0..7: the B operand of lane (li, g) for v_mfma_f32_16x16x32_bf16. The MFMA sums over
// all (g, e) of a k-group, so any bijection k(s, g, e) serves as long as A uses the same one; k = 32 s + 16 (e / 4) + 4 g +
// e % 4 makes the four lanes of a gathered row read one contiguous 64-byte half line per load instruction.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void w_split_pack_kernel(const float* __restrict__ W, bf16x8* __restrict__ Wp, int K) {
  constexpr int NB = COUT / 16, KS = CIN / 32;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= K * KS * NB * 64) return;
  const int lane = idx & 63, q = idx >> 6, nb = q % NB, s = (q / NB) % KS, o = q / (NB * KS);
  const int g = lane >> 4, li = lane & 15;
  bf16x8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = W[((int64_t)o * CIN + k_of<CIN>(s, g, e)) * COUT + nb * 16 + li];
    const __bf16 h = (__bf16)v;
    hi[e] = h;
    lo[e] = (__bf16)(v - (float)h);
  }
  bf16x8* dst = Wp + (((int64_t)o * KS + s) * NB + nb) * 128;
  dst[lane] = hi;
  dst[64 + lane] = lo;
}

// MODE (measurement builds; 1-4 give wrong results): 1 = no MFMAs, 2 = no row gathers, 3 = no W fetch / store, 4 = 2 + 3,
// 5 = row gathers with the non-temporal cache policy
template <int CIN, int COUT, int TPW, int MODE = 0, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void sparse_conv_fwd_bf16x3_kernel(
    const float* __restrict__ X, const bf16x8* __restrict__ Wp, const int* __restrict__ packed,
    const int* __restrict__ perm, float* __restrict__ Y, int n_out, int K, int ntiles,
    const unsigned* __restrict__ cmask, const int* __restrict__ cbase, unsigned x_bytes,
    const int* __restrict__ tile_order) {
  static_assert(CIN % 32 == 0 && COUT % 16 == 0, "k-groups of 32 channels, 16-column blocks");
  constexpr int NB = COUT / 16, KS = CIN / 32, ROWS = 16 * NW * TPW, NT = 64 * NW;
  constexpr int WVEC = KS * NB * 128;               // 16-byte vectors per offset (hi + lo)
  static_assert(WVEC % NT == 0, "whole vectors per thread");
  constexpr int WPT = WVEC / NT;
  constexpr int NBC = NB < 4 ? NB : 4;              // column blocks whose B fragments are held at once
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16x8* w_lds0 = reinterpret_cast<bf16x8*>(smem);
  bf16x8* w_lds1 = w_lds0 + WVEC;
  int* nbr_lds = reinterpret_cast<int*>(smem + 2 * 16 * WVEC);       // <= ROWS * K packed indices
  __shared__ unsigned wg_mask_sh[NW];
  __shared__ unsigned row_mask_sh[ROWS];
  __shared__ int row_base_sh[ROWS];

  const int pos = xcd_remap(blockIdx.x, gridDim.x);
  if (pos >= ntiles) return;
  const int tile = (tile_order && ROWS == 64) ? tile_order[pos] : pos;    // the order is over 64-row tiles
  const int row0 = tile * ROWS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  {
    const int nrow = min(ROWS, n_out - row0);
    const int b0 = cbase[row0], b1 = cbase[row0 + nrow];
    for (int t = threadIdx.x; t < ROWS; t += NT) {
      row_mask_sh[t] = t < nrow ? cmask[row0 + t] : 0u;
      row_base_sh[t] = (t < nrow ? cbase[row0 + t] : b1) - b0;
    }
    for (int t = threadIdx.x; t < b1 - b0; t += NT) nbr_lds[t] = packed[b0 + t];
  }
  int out_row[TPW][4];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int srow = row0 + (wave * TPW + j) * 16 + g * 4 + rg;
      out_row[j][rg] = srow < n_out ? (perm ? perm[srow] : srow) : -1;
    }
  __syncthreads();
  unsigned my_mask[TPW], sm[TPW];
  int my_lb[TPW];
  unsigned wsm = 0;
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    my_mask[j] = row_mask_sh[(wave * TPW + j) * 16 + li];
    my_lb[j] = row_base_sh[(wave * TPW + j) * 16 + li];
    unsigned m = my_mask[j];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m |= (unsigned)__shfl_xor((int)m, d);
    sm[j] = __builtin_amdgcn_readfirstlane(m);
    wsm |= sm[j];
  }
  if (lane == 0) wg_mask_sh[wave] = wsm;
  __syncthreads();
  unsigned todo = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) todo |= wg_mask_sh[w];

  auto nbr_of = [&](int j, int o) -> int {
    const int idx = nbr_lds[my_lb[j] + __popc(my_mask[j] & ((1u << o) - 1u))];     // in-bounds also when bit o is clear
    return ((my_mask[j] >> o) & 1u) ? idx : -1;
  };

  f32x4 acc[TPW][NB];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[j][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  bf16x8 wreg[WPT];
  auto w_fetch = [&](int o) {
    const bf16x8* src = Wp + (int64_t)o * WVEC + threadIdx.x;
#pragma unroll
    for (int i = 0; i < WPT; ++i) wreg[i] = src[NT * i];
  };
  auto w_store = [&](bf16x8* dst) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) dst[threadIdx.x + NT * i] = wreg[i];
  };
  // Gathers are buffer loads: a row without the offset gets an offset beyond the buffer -> the hardware returns zeros and
  // moves no data (no select, no traffic for absent rows, no traffic at all from a wave whose tiles lack the offset), and
  // every path through a phase still issues the same VMEM instructions, so the prefetch stays outstanding across it.
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)x_bytes, 0x00020000);
  typedef u32x4 AFrag[TPW][2 * KS];
  AFrag afA, afB;                                    // two register sets, alternating statically (no copies at the back edge)
  // CIN = 64 (ROWG): the four lanes of a quad read 64 contiguous bytes of ONE row per load instruction (instruction i: row
  // 4 (li / 4) + i of the tile, bytes 64 g + 16 (li % 4) ..), because the texture-address unit prices a wave load by the
  // cache lines its quads touch: the fragment-shaped load (a quad = 4 different rows) costs 64 line look-ups per
  // instruction, this one 16. A 4x4 transpose inside each quad (DPP, VALU that idles anyway) then hands lane (li, g) the
  // channels 16 g .. 16 g + 15 of its own row li.
  constexpr bool ROWG = CIN == 64;
  auto gather = [&](AFrag& af, int o) {
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int r = nbr_of(j, o);
      if constexpr (ROWG) {
        const int cb = 64 * g + 16 * (lane & 3);
        const int r0 = __builtin_amdgcn_update_dpp(0, r, 0x00, 0xF, 0xF, false);
        const int r1 = __builtin_amdgcn_update_dpp(0, r, 0x55, 0xF, 0xF, false);
        const int r2 = __builtin_amdgcn_update_dpp(0, r, 0xAA, 0xF, 0xF, false);
        const int r3 = __builtin_amdgcn_update_dpp(0, r, 0xFF, 0xF, 0xF, false);
        af[j][0] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, r0 < 0 ? (int)0x80000000 : r0 * 256 + cb, 0, 0);
        af[j][1] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, r1 < 0 ? (int)0x80000000 : r1 * 256 + cb, 0, 0);
        af[j][2] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, r2 < 0 ? (int)0x80000000 : r2 * 256 + cb, 0, 0);
        af[j][3] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, r3 < 0 ? (int)0x80000000 : r3 * 256 + cb, 0, 0);
      } else {
        const int voff = r < 0 ? (int)0x80000000 : r * (CIN * 4) + 16 * g;
#pragma unroll
        for (int q = 0; q < 2 * KS; ++q)
          af[j][q] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff, 64 * q, MODE == 5 ? 2 : 0);   // 5: nt policy
      }
    }
  };
  bf16x8 ah[TPW][KS], al[TPW][KS];
  auto split = [&](const AFrag& af) {
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      if constexpr (ROWG) {
        // tr[c](quad lane q) = af[q](quad lane c): two butterfly stages (lane bit 0 / register bit 0, then bit 1)
        const bool q0 = (lane & 1) != 0, q1 = (lane & 2) != 0;
        u32x4 t1[4], tr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const u32x4 other = dpp4<0xB1>(af[j][r ^ 1]);         // quad_perm [1,0,3,2]
          t1[r] = (((r & 1) != 0) == q0) ? af[j][r] : other;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const u32x4 other = dpp4<0x4E>(t1[r ^ 2]);            // quad_perm [2,3,0,1]
          tr[r] = (((r & 2) != 0) == q1) ? t1[r] : other;
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) split8(tr[2 * s], tr[2 * s + 1], ah[j][s], al[j][s]);
      } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) split8(af[j][2 * s], af[j][2 * s + 1], ah[j][s], al[j][s]);
      }
    }
  };
  // tiles of the wave that have the offset: wave-uniform bit set `act`; consecutive MFMAs go to different accumulators
  auto mfma_block = [&](const bf16x8* wl, unsigned act) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int c0 = 0; c0 < NB; c0 += NBC) {
        bf16x8 bh[NBC], bl[NBC];
#pragma unroll
        for (int c = 0; c < NBC; ++c) {
          bh[c] = wl[(s * NB + c0 + c) * 128 + lane];
          bl[c] = wl[(s * NB + c0 + c) * 128 + 64 + lane];
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int c = 0; c < NBC; ++c)
#pragma unroll
            for (int j = 0; j < TPW; ++j)
              if ((act >> j) & 1u)
                acc[j][c0 + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p == 0 ? al[j][s] : ah[j][s],
                                                                        p == 1 ? bl[c] : bh[c], acc[j][c0 + c], 0, 0, 0);
      }
    }
  };
  auto mfma_phase = [&](const bf16x8* wl, int o) {
    unsigned act = 0;
#pragma unroll
    for (int j = 0; j < TPW; ++j) act |= ((sm[j] >> o) & 1u) << j;
    if constexpr (TPW == 1) {
      if (act) mfma_block(wl, 1u);
    } else {
      if (act == 3u) mfma_block(wl, 3u);
      else if (act == 1u) mfma_block(wl, 1u);
      else if (act == 2u) mfma_block(wl, 2u);
    }
  };

  int cur = todo ? __ffs(todo) - 1 : -1;
  if (cur >= 0) {
    w_fetch(cur);
    gather(afA, cur);
    w_store(w_lds0);
  }
  __syncthreads();
  // one phase: issue the NEXT offset's W fetch and row gather first, then split the rows gathered one phase ago, multiply,
  // hand W over. Unrolled by two so that the two row sets and the two W buffers alternate statically.
  auto phase = [&](AFrag& a_cur, AFrag& a_nxt, const bf16x8* wl_cur, bf16x8* wl_nxt) {
    todo &= todo - 1;
    const int nxt = todo ? __ffs(todo) - 1 : -1;
    const int oq = nxt >= 0 ? nxt : cur;             // the last phase re-fetches its own offset (unused)
    if constexpr (MODE != 3 && MODE != 4) w_fetch(oq);
    if constexpr (MODE != 2 && MODE != 4) gather(a_nxt, oq);
    split(a_cur);
    if constexpr (MODE != 1) {
      mfma_phase(wl_cur, cur);
    } else {                                         // keep the gathered rows and the split alive
#pragma unroll
      for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int s = 0; s < KS; ++s) acc[j][0][0] += (float)ah[j][s][0] + (float)al[j][s][7] + (float)ah[j][s][4];
    }
    if constexpr (MODE != 3 && MODE != 4) w_store(wl_nxt);
    __syncthreads();
    cur = nxt;
  };
  while (cur >= 0) {
    phase(afA, afB, w_lds0, w_lds1);
    if (cur < 0) break;
    phase(afB, afA, w_lds1, w_lds0);
  }

#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
      if (out_row[j][rg] >= 0) {
        float* dst = Y + (int64_t)out_row[j][rg] * COUT + li;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) dst[nb * 16] = acc[j][nb][rg];
      }
}

template <int CIN, int COUT, int TPW, int NW = 4>
int launch_tpw(const float* X, const bf16x8* Wp, const unsigned* cmask, const int* cbase, const int* packed, const int* perm,
               const int* tile_order, float* Y, int64_t n_in, int64_t n_out, int K, hipStream_t st) {
  constexpr int WVEC = (CIN / 32) * (COUT / 16) * 128;
  const int ntiles = crb_cdiv(n_out, 16 * NW * TPW);
  const int grid = ((ntiles + 7) / 8) * 8;
  const size_t lds = 2 * 16 * (size_t)WVEC + sizeof(int) * 16 * NW * TPW * K;
  auto kern = sparse_conv_fwd_bf16x3_kernel<CIN, COUT, TPW, 0, NW>;
  if constexpr (CIN == 64 && COUT == 64 && NW == 4) {
    if (g_bf16x3_mode == 1) kern = sparse_conv_fwd_bf16x3_kernel<CIN, COUT, TPW, 1>;
    if (g_bf16x3_mode == 2) kern = sparse_conv_fwd_bf16x3_kernel<CIN, COUT, TPW, 2>;
    if (g_bf16x3_mode == 3) kern = sparse_conv_fwd_bf16x3_kernel<CIN, COUT, TPW, 3>;
    if (g_bf16x3_mode == 4) kern = sparse_conv_fwd_bf16x3_kernel<CIN, COUT, TPW, 4>;
    if (g_bf16x3_mode == 5) kern = sparse_conv_fwd_bf16x3_kernel<CIN, COUT, TPW, 5>;
  }
  static bool attr_done = false;                    // per instantiation; the measurement builds set it every time
  if (!attr_done || g_bf16x3_mode) {
    CRB_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, st, X, Wp, packed, perm, Y, (int)n_out, K, ntiles, cmask, cbase,
                     (unsigned)(n_in * CIN * 4), tile_order);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

template <int CIN, int COUT>
int launch_bf16x3(const float* X, const float* W, const unsigned* cmask, const int* cbase, const int* packed, const int* perm,
                  const int* tile_order, float* Y, int64_t n_in, int64_t n_out, int K, void* ws, int64_t wsb, hipStream_t st) {
  if (n_in * CIN * 4 >= (int64_t)1 << 31) return CRB_ERR_UNSUPPORTED;      // 32-bit buffer offsets, top half = "absent row"
  if (!ws || wsb < (int64_t)K * CIN * COUT * 4) return CRB_ERR_WORKSPACE;
  bf16x8* Wp = (bf16x8*)ws;
  const int total = K * (CIN / 32) * (COUT / 16) * 64;
  hipLaunchKernelGGL((w_split_pack_kernel<CIN, COUT>), dim3(crb_cdiv(total, 256)), dim3(256), 0, st, W, Wp, K);
  constexpr bool BIG = CIN * COUT > 64 * 64;       // two tiles per wave would spill there
  // measured on the SECOND bs=16 tables (tools/bench_sparse_conv.py): one tile per wave, 4 waves 85-103 us (L3) / 60 us (L4);
  // two tiles per wave 112-118 / 78; 8 waves x one tile 106 / 72
  const int tpw = g_bf16x3_tpw ? g_bf16x3_tpw : 1;
  if constexpr (!BIG && ((CIN / 32) * (COUT / 16) * 128) % 512 == 0) {
    if (tpw == 3) return launch_tpw<CIN, COUT, 1, 8>(X, Wp, cmask, cbase, packed, perm, tile_order, Y, n_in, n_out, K, st);
  }
  if (BIG || tpw == 1) return launch_tpw<CIN, COUT, 1>(X, Wp, cmask, cbase, packed, perm, tile_order, Y, n_in, n_out, K, st);
  if constexpr (!BIG) return launch_tpw<CIN, COUT, 2>(X, Wp, cmask, cbase, packed, perm, tile_order, Y, n_in, n_out, K, st);
  return CRB_ERR_UNSUPPORTED;
}

}  // namespace

#define CRB_BF16X3_SHAPES(X_) X_(32, 32) X_(32, 64) X_(64, 32) X_(64, 64) X_(64, 128) X_(128, 64)

extern "C" int crb_sparse_conv_bf16x3_supported(int cin, int cout) {
#define X_(a, b) if (cin == a && cout == b) return 1;
  CRB_BF16X3_SHAPES(X_)
#undef X_
  return 0;
}

extern "C" int64_t crb_sparse_conv_bf16x3_workspace_bytes(int K, int cin, int cout) {
  return (int64_t)K * cin * cout * 4;
}

#ifdef CRB_MEASURE
extern "C" int crb_sparse_conv_bf16x3_set_mode(int mode) {
  g_bf16x3_mode = (mode >= 1 && mode <= 5) ? mode : 0;
  return CRB_OK;
}

extern "C" int crb_sparse_conv_bf16x3_set_tiles_per_wave(int tpw) {
  g_bf16x3_tpw = (tpw >= 1 && tpw <= 3) ? tpw : 0;       // 3 = one tile per wave, 8 waves per workgroup
  return CRB_OK;
}
#endif

extern "C" int crb_sparse_conv_forward_bf16x3(const float* X, const float* W, const uint32_t* cmask, const int32_t* cbase,
                                              const int32_t* packed, const int32_t* perm, const int32_t* tile_order,
                                              float* Y, int64_t n_in,
                                              int64_t n_out, int K, int cin, int cout, void* workspace,
                                              int64_t workspace_bytes, void* stream) {
  if (n_in < 0 || n_out < 0 || K <= 0 || K > 32) return CRB_ERR_ARG;
  if (n_out == 0) return CRB_OK;
  hipStream_t st = (hipStream_t)stream;
#define X_(a, b) \
  if (cin == a && cout == b) \
    return launch_bf16x3<a, b>(X, W, cmask, cbase, packed, perm, tile_order, Y, n_in, n_out, K, workspace, workspace_bytes, st);
  CRB_BF16X3_SHAPES(X_)
#undef X_
  return CRB_ERR_UNSUPPORTED;
}

#endif  // CRB_MEASURE
