// Training-time set abstraction for gfx950: the two-layer shared MLP of one StackSAModuleMSG scale
//   group -> Conv(1x1) -> BatchNorm -> ReLU -> Conv(1x1) -> BatchNorm -> ReLU -> max over the neighbourhood
// (pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py:90-108; RoI-grid pooling: pcdet/models/roi_heads/pvrcnn_head.py:102-113)
// WITHOUT the (M*ns, H) activations in HBM. At the RoI-grid shape of PV-RCNN (M = 16 x 128 x 216 queries, ns = 16, H = 64) the
// module path saved six (7.08 M, 64) f32 tensors (10.8 GB) and made ~17 passes over tensors of that size.
//
// Train-mode BatchNorm needs full-tensor statistics before the next layer can run, so the forward is three RECOMPUTE passes
// over the ball-query indices, none of which writes an activation:
//   pass 0  (crb_group_affine_rows_stats_stack, out = NULL)   y1 = W1x (xyz_j - c_i) + P[j]      -> slab sums of y1, y1^2
//   pass A  (MODE 0)   z1 = relu(bn1(y1)), y2 = z1 W2^T on the f32 MFMA                          -> per-wave sums of y2, y2^2
//   pass B  (MODE 1)   the same, then z2 = relu(bn2(y2)), max over the ns samples                -> out (M,H2), arg, y2 at the arg
// (P = F W1f^T is one small GEMM over the N source points, as in the inference kernel sa_mlp.hip.) The backward is
//   sums    (crb_bn_relu_max_backward_sums)  dbeta2, dgamma2 from the M x H2 selected entries
//   pass C  (MODE 2)   recompute y1, z1, y2; dy2 = BatchNorm backward of the max's scatter (in registers);
//                      dz1 = dy2 W2 (MFMA) masked by z1 > 0 -> written ONCE as (M*ns, H1);  dW2 += dy2^T z1 (MFMA, rows = k);
//                      per-wave sums for BatchNorm 1's backward (sum d, sum d xhat)
//   pass D  (crb_group_affine_rows_grad_bn_recompute_stack)  BatchNorm 1 backward applied to that tensor while it is loaded,
//                      y1 recomputed from P, scatter-add into dP (pointnet2_stack.hip)
// Why not recompute in pass D as well (the review's two-pass proposal): an f32 64x64 GEMM pass over 7.08 M rows costs ~0.6 ms on
// the MFMA (58 GF at ~0.6 of 157 TF), writing + reading one (7.08 M, 64) tensor ~0.7 ms, and D would need two GEMMs.
//
// Lane layout (one wave = one query's 16-sample tile; lane = (r, g), r = lane & 15 = sample, g = lane >> 4):
//   a lane holds channels {16 b + 4 g + j} (b = block, j = 0..3) of ITS sample for both layers. With the contraction index of
//   MFMA step (b, j) chosen as {16 b + 4 g' + j : g' = 0..3}, z1 in that layout IS the B operand of y2^T = W2 z1^T, y2^T comes
//   out in the same layout (accumulator register j of block b = channel 16 b + 4 g + j, column = sample r), and dy2 in that
//   layout IS the B operand of dz1^T = W2^T dy2^T: no transposes between the layers. Only dW2 (contraction over samples)
//   needs the tiles through LDS (wave-private, no barrier). W2 operands are read from LDS images (one ds_read per MFMA step).
// Numerics: y1 and the BatchNorm expressions are formed exactly like group_affine_rows_kernel / bn_apply_kernel form them
// (fmaf order, ga * ((y - mu) * is) + be, no contraction); the GEMMs sum in MFMA order. Empty balls are ordinary rows with a zero
// grouped row (the reference zeroes them: pointnet2_utils.py:141), they count in every statistic. Everything is deterministic:
// per-wave / per-workgroup partials, reduced in index order.
#include "crb_common.h"
#include "../../include/crb_hip.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

struct SaTrainArgs {
  int B;
  int64_t M;
  int ns;
  const float* xyz;
  const int* xyz_cnt;
  const float* P;          // (N, H1)
  const float* new_xyz;
  const int* new_cnt;
  const int* idx;          // (M, ns)
  const unsigned char* empty;
  const float* W1x;        // (3, H1)
  const float* W2;         // (H2, H1)
  const float* bn1[4];     // mean, invstd, gamma, beta (H1)
  const float* bn2[4];     // (H2); MODE 0: unused
  // MODE 0
  float* stat;             // (waves, 2, H2)
  // MODE 1
  float* out;              // (M, ld_out) view
  int64_t ld_out;
  int* arg;                // (M, H2)
  float* ysel;             // (M, H2): y2 at the arg sample
  // MODE 2
  const float* gout;       // (M, ld_g) view
  int64_t ld_g;
  const float* dbeta2;
  const float* dgamma2;
  float inv_n;
  float* gz1;              // (M*ns, H1): dz1 [z1 > 0]
  float* part1;            // (waves, 2, H1): sums of d, d * xhat1
  float* partW;            // (blocks, H2, H1)
  int skip;                // measurement builds (wrong results): bit 0 = no MFMAs of the forward GEMM, bit 1 = every P row is row 0
                           // (no gather misses), bit 2 = no layer 1 (z1 = the P slice); 0 in the product library
};
#ifdef CRB_MEASURE
int g_sat_skip = 0;
#else
constexpr int g_sat_skip = 0;
#endif

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// reductions over the 16 lanes of a DPP row (= the 16 samples of a tile); every lane ends with the result
__device__ __forceinline__ float row_max16(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));      // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_f<0x4E>(v));      // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_f<0x141>(v));     // row_half_mirror
  v = fmaxf(v, dpp_f<0x140>(v));     // row_mirror
  return v;
}
// the same for NON-NEGATIVE finite floats (relu outputs): their bit patterns order like integers, and an integer maximum needs no
// NaN canonicalisation of its operands (fmaxf costs a v_max v, v, v per step on top of the DPP move)
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ float row_max16_nonneg(float f) {
  int v = __builtin_bit_cast(int, f);
  v = max(v, dpp_i<0xB1>(v));
  v = max(v, dpp_i<0x4E>(v));
  v = max(v, dpp_i<0x141>(v));
  v = max(v, dpp_i<0x140>(v));
  return __builtin_bit_cast(float, v);
}
__device__ __forceinline__ float row_sum16(float v) {
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);
  v += dpp_f<0x140>(v);
  return v;
}

template <int H1, int H2, int MODE>
struct SaLds {
  static constexpr int W2A = 0;                                   // [H1 steps][64 lanes][NB]
  static constexpr int W2B = W2A + H1 * H2;                       // MODE 2: [H2 steps][64 lanes][MB]
  static constexpr int PAR = W2B + (MODE == 2 ? H1 * H2 : 0);     // w1x 3 H1 | bn1 4 H1 | bn2 4 H2 | db2 dg2 2 H2
  static constexpr int PAR_FLOATS = 7 * H1 + 6 * H2;
  static constexpr int HP1 = H1 + 4, HP2 = H2 + 4;
  static constexpr int TR = PAR + PAR_FLOATS;                     // MODE 2: per wave [16][HP2] dy2 | [16][HP1] z1
  static constexpr int TR_WAVE = 16 * (HP1 + HP2);
  static constexpr int CON = TR + (MODE == 2 ? 4 * TR_WAVE : 0);  // MODE >= 1: the two constant rows of an empty ball (y2, relu(bn2(y2)))
  static constexpr int VS = CON + (MODE >= 1 ? 2 * H2 : 0);       // MODE 2: vsum of every wave
  static constexpr int TOTAL = VS + (MODE == 2 ? 4 * H2 : 0);
};

template <int N>
__device__ __forceinline__ void lds_read_vec(const float* p, float (&v)[N]) {
  if constexpr (N == 4) {
    const f4 t = *reinterpret_cast<const f4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  } else if constexpr (N == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  } else {
    v[0] = p[0];
  }
}

template <int H1, int H2, int MODE>
__global__ __launch_bounds__(256, MODE == 0 ? 4 : MODE == 1 ? 3 : 2) void sa_train_kernel(SaTrainArgs a) {
  constexpr int MB = H1 / 16, NB = H2 / 16;
  using L = SaLds<H1, H2, MODE>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW2A = smem + L::W2A;
  float* sW2B = smem + L::W2B;
  float* sW1x = smem + L::PAR;
  float* sBn1 = sW1x + 3 * H1;
  float* sBn2 = sBn1 + 4 * H1;
  float* sD2 = sBn2 + 4 * H2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;

  // ---- operand images and parameters
  for (int e = threadIdx.x; e < H1 * H2; e += 256) {
    {
      const int nb = e % NB, ln = (e / NB) & 63, st = e / (NB * 64);
      const int mb = st >> 2, j = st & 3, rr = ln & 15, gg = ln >> 4;
      sW2A[e] = a.W2[(16 * nb + rr) * H1 + 16 * mb + 4 * gg + j];
    }
    if constexpr (MODE == 2) {
      const int mb = e % MB, ln = (e / MB) & 63, st = e / (MB * 64);
      const int nb = st >> 2, j = st & 3, rr = ln & 15, gg = ln >> 4;
      sW2B[e] = a.W2[(16 * nb + 4 * gg + j) * H1 + 16 * mb + rr];
    }
  }
  for (int e = threadIdx.x; e < 3 * H1; e += 256) sW1x[e] = a.W1x[e];
  for (int e = threadIdx.x; e < 4 * H1; e += 256) sBn1[e] = a.bn1[e / H1][e % H1];
  if constexpr (MODE >= 1)
    for (int e = threadIdx.x; e < 4 * H2; e += 256) sBn2[e] = a.bn2[e / H2][e % H2];
  if constexpr (MODE == 2)
    for (int e = threadIdx.x; e < 2 * H2; e += 256) sD2[e] = (e < H2 ? a.dbeta2[e] : a.dgamma2[e - H2]) * a.inv_n;
  __syncthreads();

  const int ns = a.ns, T = ns >> 4;                    // ns is a multiple of 16 (host check)
  const int M = (int)a.M, nw = (int)gridDim.x * 4, w0 = (int)blockIdx.x * 4 + wave;     // M * ns < 2^31 (host check)

  // Work distribution: chunks of 16 consecutive queries, chunk c of wave w0 = w0 + k nw (round-robin over the waves of the launch).
  // A wave looks at FOUR chunks at a time (a "group"): one coalesced byte load gives the 64 empty flags, two ballots give the live
  // and the empty queries as bit masks. Only live queries enter the tile pipeline; an empty ball costs a set bit in a mask (forward:
  // its constant rows, backward: one row of gout), not a pipeline iteration with an exposed load (first version: 0.6 ms of the
  // RoI-grid passes went into iterations over the 84 % empty queries at r = 0.8).
  // Tile pipeline over the live queries (in ascending order): the cursor two tiles ahead ISSUES the loads of the sample index and
  // the query centre (stage A: nothing it loads is looked at in the same iteration - a wait there would also wait for the P rows
  // just requested), one tile ahead the source row is formed and the P row slice and the point are requested (stage B). The query
  // index is wave-uniform (scalar register); the frame of the cursor advances with it (first source row of the frame: `start`).
  struct Cur { int q, t, b, fend, start; };
  struct StA { int idxv, start; float cx, cy, cz; bool valid; };
  struct StB { f4 p[MB]; float px, py, pz, cx, cy, cz; int row; };   // the offset px - cx is formed where it is used
  auto load_a = [&](Cur& c) {
    StA s;
    s.idxv = 0; s.start = 0; s.cx = s.cy = s.cz = 0.f;
    const int q = __builtin_amdgcn_readfirstlane(c.q);
    s.valid = q >= 0;
    if (s.valid) {
      int bb = __builtin_amdgcn_readfirstlane(c.b), fe = __builtin_amdgcn_readfirstlane(c.fend),
          st = __builtin_amdgcn_readfirstlane(c.start);
      while (q >= fe && bb + 1 < a.B) {
        st += a.xyz_cnt[bb];
        ++bb;
        fe += a.new_cnt[bb];
      }
      c.b = bb; c.fend = fe; c.start = st;
      s.start = st;
      s.idxv = a.idx[(int64_t)q * ns + 16 * c.t + r];
      s.cx = a.new_xyz[(int64_t)q * 3 + 0]; s.cy = a.new_xyz[(int64_t)q * 3 + 1]; s.cz = a.new_xyz[(int64_t)q * 3 + 2];
    }
    return s;
  };
  auto load_b = [&](const StA& s) {
    StB b;
    // (only live queries get here; past the end of the wave's work the cursor is invalid and row 0 is read and never used: no
    // branch around the loads, no zero-initialised operand set)
    b.row = s.valid ? s.start + s.idxv : 0;
    b.cx = s.cx; b.cy = s.cy; b.cz = s.cz;
    const float* src = a.P + (int64_t)((a.skip & 2) ? 0 : b.row) * H1 + 4 * g;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) b.p[mb] = *reinterpret_cast<const f4*>(src + 16 * mb);
    b.px = a.xyz[(int64_t)b.row * 3 + 0];
    b.py = a.xyz[(int64_t)b.row * 3 + 1];
    b.pz = a.xyz[(int64_t)b.row * 3 + 2];
    return b;
  };

  // ---- the tile's arithmetic, shared by the modes
  // layer 1: y1 -> xhat1 -> z1 in the (sample r, channels 16 mb + 4 g + j) layout; live = false: the zero grouped row of an empty ball
  auto layer1 = [&](const StB& b, bool live, f4 (&z1)[MB], f4 (&xh1)[MB]) {
    if (a.skip & 4) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) z1[mb] = xh1[mb] = b.p[mb];
      return;
    }
    const float dx = b.px - b.cx, dy = b.py - b.cy, dz = b.pz - b.cz;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int c = 16 * mb + 4 * g;
      const f4 w0v = *reinterpret_cast<const f4*>(sW1x + c), w1v = *reinterpret_cast<const f4*>(sW1x + H1 + c),
               w2v = *reinterpret_cast<const f4*>(sW1x + 2 * H1 + c);
      f4 y = f4{0.f, 0.f, 0.f, 0.f};
      if (live) {
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = fmaf(w2v[k], dz, fmaf(w1v[k], dy, fmaf(w0v[k], dx, b.p[mb][k])));
      }
      const f4 mu = *reinterpret_cast<const f4*>(sBn1 + c), is = *reinterpret_cast<const f4*>(sBn1 + H1 + c),
               ga = *reinterpret_cast<const f4*>(sBn1 + 2 * H1 + c), be = *reinterpret_cast<const f4*>(sBn1 + 3 * H1 + c);
      xh1[mb] = (y - mu) * is;
      f4 z = ga * xh1[mb] + be;
#pragma unroll
      for (int k = 0; k < 4; ++k) z[k] = z[k] > 0.f ? z[k] : 0.f;
      z1[mb] = z;
    }
  };
  // y2^T = W2 z1^T
  auto gemm2 = [&](const f4 (&z1)[MB], f4 (&acc)[NB]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = f4{0.f, 0.f, 0.f, 0.f};
    if (a.skip & 1) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = z1[nb % MB];
      return;
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float wa[NB];
        lds_read_vec<NB>(sW2A + ((mb * 4 + j) * 64 + lane) * NB, wa);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[nb], z1[mb][j], acc[nb], 0, 0, 0);
      }
  };
  auto bn2 = [&](const f4 (&acc)[NB], f4 (&xh2)[NB], f4 (&z2)[NB]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int c = 16 * nb + 4 * g;
      const f4 mu = *reinterpret_cast<const f4*>(sBn2 + c), is = *reinterpret_cast<const f4*>(sBn2 + H2 + c),
               ga = *reinterpret_cast<const f4*>(sBn2 + 2 * H2 + c), be = *reinterpret_cast<const f4*>(sBn2 + 3 * H2 + c);
      xh2[nb] = (acc[nb] - mu) * is;
      z2[nb] = ga * xh2[nb] + be;
    }
  };

  // running state
  f4 s1[NB], s2[NB];                       // MODE 0
  f4 runz[NB], ybest[NB];                  // MODE 1 (row-uniform)
  i4 abest[NB];
  f4 sdb[MB], sdg[MB];                     // MODE 2
  f4 accW[NB][MB];
  float vs = 0.f, k2pos = 0.f;             // MODE 2, lane = channel: sum over this wave's EMPTY queries of k2 gout [z2 > 0]; k2 [z2 > 0]
  int64_t n_empty = 0;                     // empty queries of this wave
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    s1[nb] = s2[nb] = f4{0.f, 0.f, 0.f, 0.f};
    runz[nb] = f4{-1.f, -1.f, -1.f, -1.f};
    ybest[nb] = f4{0.f, 0.f, 0.f, 0.f};
    abest[nb] = i4{0, 0, 0, 0};
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) accW[nb][mb] = f4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) sdb[mb] = sdg[mb] = f4{0.f, 0.f, 0.f, 0.f};
  float* trD = smem + L::TR + wave * L::TR_WAVE;       // [16][HP2]
  float* trZ = trD + 16 * L::HP2;                      // [16][HP1]
  float* sCon = smem + L::CON;                         // MODE >= 1: [y2 of an empty ball's rows (H2)][relu(bn2(.)) of it (H2)]
  float* sVsum = smem + L::VS + wave * H2;             // MODE 2

  // MODE 2, one tile: dz1^T = W2^T dy2^T (dy2 in the accumulator layout IS the B operand), masked by z1 > 0 -> gz1 row (optional),
  // the sums of BatchNorm 1's backward, and dW2 += dy2^T z1 with both tiles read back from LDS with the samples as k
  auto bwd_tile = [&](const f4 (&dy2)[NB], const f4 (&z1)[MB], const f4 (&xh1)[MB], float* grow) {
    if constexpr (MODE == 2) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) *reinterpret_cast<f4*>(trD + r * L::HP2 + 16 * nb + 4 * g) = dy2[nb];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) *reinterpret_cast<f4*>(trZ + r * L::HP1 + 16 * mb + 4 * g) = z1[mb];
      f4 acc1[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) acc1[mb] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float wb[MB];
          lds_read_vec<MB>(sW2B + ((nb * 4 + j) * 64 + lane) * MB, wb);
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
            acc1[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[mb], dy2[nb][j], acc1[mb], 0, 0, 0);
        }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        f4 d;
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = z1[mb][k] > 0.f ? acc1[mb][k] : 0.f;
        if (grow) *reinterpret_cast<f4*>(grow + 16 * mb) = d;
        sdb[mb] += d;
        sdg[mb] += d * xh1[mb];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        float av[NB], bv[MB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) av[nb] = trD[(4 * tt + g) * L::HP2 + 16 * nb + r];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) bv[mb] = trZ[(4 * tt + g) * L::HP1 + 16 * mb + r];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
            accW[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[nb], bv[mb], accW[nb][mb], 0, 0, 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  };

  // An empty ball's 16 (or 32) rows are equal: z1, y2, z2 are per-launch constants. The forward modes never run them through
  // the MFMA; MODE 2 reduces their whole contribution to one H2-vector per wave (below).
  StB b_zero;
  b_zero.row = -1; b_zero.px = b_zero.py = b_zero.pz = b_zero.cx = b_zero.cy = b_zero.cz = 0.f;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) b_zero.p[mb] = f4{0.f, 0.f, 0.f, 0.f};
  if constexpr (MODE >= 1) {
    f4 z1c[MB], xh1c[MB], accc[NB], xh2c[NB], z2c[NB];
    layer1(b_zero, false, z1c, xh1c);
    gemm2(z1c, accc);
    bn2(accc, xh2c, z2c);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (wave == 0 && r == 0) {
          sCon[16 * nb + 4 * g + j] = accc[nb][j];
          sCon[H2 + 16 * nb + 4 * g + j] = z2c[nb][j] > 0.f ? z2c[nb][j] : 0.f;
        }
      }
    __syncthreads();
    if constexpr (MODE == 2) {
      if (lane < H2) k2pos = sCon[H2 + lane] > 0.f ? sBn2[2 * H2 + lane] * sBn2[H2 + lane] : 0.f;      // gamma * invstd where relu' = 1
    }
  }

  // ---- the live-query iterator
  auto load_flags = [&](int gq) -> int {
    const int c = w0 + (4 * gq + (lane >> 4)) * nw;
    const int64_t q = (int64_t)c * 16 + (lane & 15);
    return q < M ? (int)a.empty[q] : 2;
  };
  auto query_of = [&](int gq, int bit) { return 16 * (w0 + (4 * gq + (bit >> 4)) * nw) + (bit & 15); };
  int grp = 0;
  int f_cur = load_flags(0), f_nxt = load_flags(1);
  unsigned long long live_bits = 0;
  auto enter_group = [&]() {               // f_cur = the flags of group grp
    live_bits = __ballot(f_cur == 0);
    unsigned long long emp = __ballot(f_cur == 1);
    n_empty += __popcll(emp);
    if constexpr (MODE >= 1) {
      while (emp) {
        const int q = query_of(grp, __builtin_ctzll(emp));
        emp &= emp - 1;
        if (lane < H2) {
          if constexpr (MODE == 1) {       // the constant rows of an empty ball: 64 lanes, one row each of out / arg / y_sel
            a.out[(int64_t)q * a.ld_out + lane] = sCon[H2 + lane];
            a.arg[(int64_t)q * H2 + lane] = 0;
            a.ysel[(int64_t)q * H2 + lane] = sCon[lane];
          } else {                         // the max of equal rows sits at sample 0 (arg = 0): the only row with a gradient
            vs += k2pos * a.gout[(int64_t)q * a.ld_g + lane];
          }
        }
      }
    }
  };
  enter_group();
  auto next_live = [&]() -> int {
    while (live_bits == 0) {
      if ((int64_t)(w0 + 4 * (grp + 1) * nw) * 16 >= M) return -1;       // the next group's first chunk is past the end
      ++grp;
      f_cur = f_nxt;
      f_nxt = load_flags(grp + 1);
      enter_group();
    }
    const int bit = __builtin_ctzll(live_bits);
    live_bits &= live_bits - 1;
    return query_of(grp, bit);
  };
  auto advance = [&](Cur& c) {
    if (c.q >= 0 && ++c.t == T) { c.t = 0; c.q = next_live(); }
  };

  // one tile: query q, row tile t, operands b_cur
  auto compute = [&](const StB& b_cur, const int q, const int t) {
      // MODE 1 carries 48 registers of running maxima: keep the loop-invariant LDS reads (operand image, parameters) inside the
      // loop instead of hoisted into registers (78 spills otherwise); MODE 0: the registers go to the second prefetch stage
      if constexpr (MODE == 1) asm volatile("" ::: "memory");
      f4 z1[MB], xh1[MB], acc[NB];
      layer1(b_cur, true, z1, xh1);
      gemm2(z1, acc);
      if constexpr (MODE == 0) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          s1[nb] += acc[nb];
          s2[nb] += acc[nb] * acc[nb];
        }
      } else {
        const int smp = 16 * t + r;
        f4 xh2[NB], z2[NB];
        bn2(acc, xh2, z2);
        if constexpr (MODE == 1) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float z = z2[nb][j] > 0.f ? z2[nb][j] : 0.f;
              const float zm = row_max16_nonneg(z);
              const unsigned long long bal = __ballot(z == zm);
              const unsigned rowbits = (unsigned)(bal >> (16 * g)) & 0xffffu;
              const int first = __ffs((int)rowbits) - 1;                 // lowest sample of the tile that attains the maximum
              const float yv = acc[nb][j];
              const float yw = __shfl(yv, 16 * g + first, 64);           // y2 of that sample
              if (zm > runz[nb][j]) {                                    // strict: an earlier tile keeps a tie
                runz[nb][j] = zm;
                ybest[nb][j] = yw;
                abest[nb][j] = 16 * t + first;
              }
            }
          if (t == T - 1) {
            if (r == 0) {
#pragma unroll
              for (int nb = 0; nb < NB; ++nb) {
                const int c = 16 * nb + 4 * g;
                *reinterpret_cast<f4*>(a.out + (int64_t)q * a.ld_out + c) = runz[nb];
                *reinterpret_cast<i4*>(a.arg + (int64_t)q * H2 + c) = abest[nb];
                *reinterpret_cast<f4*>(a.ysel + (int64_t)q * H2 + c) = ybest[nb];
              }
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) runz[nb] = f4{-1.f, -1.f, -1.f, -1.f};
          }
        } else {
          // ---- dy2 = BatchNorm-2 backward of the max's scatter, in registers
          f4 dy2[NB];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const int c = 16 * nb + 4 * g;
            const f4 gs = *reinterpret_cast<const f4*>(a.gout + (int64_t)q * a.ld_g + c);
            const i4 av = *reinterpret_cast<const i4*>(a.arg + (int64_t)q * H2 + c);
            const f4 is = *reinterpret_cast<const f4*>(sBn2 + H2 + c), ga = *reinterpret_cast<const f4*>(sBn2 + 2 * H2 + c);
            const f4 db = *reinterpret_cast<const f4*>(sD2 + c), dg = *reinterpret_cast<const f4*>(sD2 + H2 + c);
            f4 d;
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = (z2[nb][k] > 0.f && av[k] == smp) ? gs[k] : 0.f;
            dy2[nb] = ga * is * (d - db - xh2[nb] * dg);
          }
          float* grow = a.gz1 + ((int64_t)q * ns + smp) * H1 + 4 * g;
          bwd_tile(dy2, z1, xh1, grow);
        }
      }
  };

  Cur cur{next_live(), 0, 0, a.new_cnt[0], 0};
  int q = cur.q, t = 0;                    // the tile being computed
  StA a_nxt = load_a(cur);
  // one tile of look-ahead for the P rows (two tiles, with three operand sets in rotation, measured the same: the gathers are not
  // what a tile waits for - profiles/r05_sa_train_skip_work.txt)
  StB b_cur = load_b(a_nxt);
  advance(cur);
  int q1 = cur.q, t1 = cur.t;
  a_nxt = load_a(cur);
  // (two operand sets used alternately with the loop unrolled by two - no copies - measured slower: the doubled body spills)
  while (q >= 0) {
    StB b_nxt = load_b(a_nxt);
    advance(cur);
    const int q2 = cur.q, t2 = cur.t;
    a_nxt = load_a(cur);
    compute(b_cur, q, t);
    b_cur = b_nxt;
    q = q1; t = t1;
    q1 = q2; t1 = t2;
  }

  const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
  if constexpr (MODE == 0) {
    f4 z1c[MB], xh1c[MB], accc[NB];
    layer1(b_zero, false, z1c, xh1c);
    gemm2(z1c, accc);
    const float ne = (float)(n_empty * T);  // empty tiles of this wave; every lane is one of the 16 equal rows of each
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u = row_sum16(s1[nb][j] + ne * accc[nb][j]), v = row_sum16(s2[nb][j] + ne * (accc[nb][j] * accc[nb][j]));
        if (r == 0) {
          a.stat[(wid * 2 + 0) * H2 + 16 * nb + 4 * g + j] = u;
          a.stat[(wid * 2 + 1) * H2 + 16 * nb + 4 * g + j] = v;
        }
      }
  }
  if constexpr (MODE == 2) {
    // the empty queries of this wave as ONE synthetic tile: with z1, xhat1, xhat2 constant on their rows, dy2 of row s is
    // base + [s = 0] k2 gout [z2 > 0], base = k2 (0 - dbeta2/n - xhat2 dgamma2/n), and everything downstream (the sums of
    // BatchNorm 1, dW2) is linear in the sum over those rows: row 0 of the tile carries n_empty ns base + vsum, rows 1..15 zero
    // (their first-layer gradient is not needed: an empty ball scatters nothing and has rel = 0).
    {
      f4 z1c[MB], xh1c[MB], accc[NB], xh2c[NB], z2c[NB], dy2[NB];
      layer1(b_zero, false, z1c, xh1c);
      gemm2(z1c, accc);
      bn2(accc, xh2c, z2c);
      if (lane < H2) sVsum[lane] = vs;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const float cnt = (float)n_empty * (float)ns;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int c = 16 * nb + 4 * g;
        const f4 is = *reinterpret_cast<const f4*>(sBn2 + H2 + c), ga = *reinterpret_cast<const f4*>(sBn2 + 2 * H2 + c);
        const f4 db = *reinterpret_cast<const f4*>(sD2 + c), dg = *reinterpret_cast<const f4*>(sD2 + H2 + c);
        const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
        const f4 base = ga * is * (zero - db - xh2c[nb] * dg);
        dy2[nb] = r == 0 ? cnt * base + *reinterpret_cast<const f4*>(sVsum + c) : zero;
      }
      if (r != 0) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) z1c[mb] = xh1c[mb] = f4{0.f, 0.f, 0.f, 0.f};
      }
      bwd_tile(dy2, z1c, xh1c, nullptr);
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u = row_sum16(sdb[mb][j]), v = row_sum16(sdg[mb][j]);
        if (r == 0) {
          a.part1[(wid * 2 + 0) * H1 + 16 * mb + 4 * g + j] = u;
          a.part1[(wid * 2 + 1) * H1 + 16 * mb + 4 * g + j] = v;
        }
      }
    // dW2 of the workgroup: the four waves add into one LDS image in wave order (accW[nb][mb][j] = dW2[16 nb + 4 g + j][16 mb + r])
    __syncthreads();
    float* sAcc = smem;                     // H2 x H1, over the operand images (no longer needed)
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float* p = sAcc + (16 * nb + 4 * g + j) * H1 + 16 * mb + r;
              *p = w == 0 ? accW[nb][mb][j] : *p + accW[nb][mb][j];
            }
      }
      __syncthreads();
    }
    float* dst = a.partW + (int64_t)blockIdx.x * H1 * H2;
    for (int e = threadIdx.x; e < H1 * H2; e += 256) dst[e] = sAcc[e];
  }
}

// out[e] = sum over k of part[k][e] in double: 16 slices of the part range per column, each in index order, then the slices in
// order. Two jobs per launch (the BatchNorm-1 sums and dW2 of one backward): blocks [0, blocksA) serve job A, the rest job B.
struct SaReduceJob { const float* part; int64_t nparts; int count; float* out; };
__global__ __launch_bounds__(256) void sa_reduce_parts_kernel(SaReduceJob ja, int blocksA, SaReduceJob jb) {
  __shared__ double red[16][16];
  const bool first = (int)blockIdx.x < blocksA;
  const SaReduceJob j = first ? ja : jb;
  const int blk = first ? blockIdx.x : blockIdx.x - blocksA;
  const int c = threadIdx.x & 15, s = threadIdx.x >> 4;
  const int e = blk * 16 + c;
  const int64_t per = (j.nparts + 15) / 16, k0 = s * per, k1 = k0 + per < j.nparts ? k0 + per : j.nparts;
  double acc = 0.0;
  if (e < j.count)
    for (int64_t k = k0; k < k1; ++k) acc += (double)j.part[k * j.count + e];
  red[s][c] = acc;
  __syncthreads();
  if (s == 0 && e < j.count) {
    for (int k = 1; k < 16; ++k) acc += red[k][c];
    j.out[e] = (float)acc;
  }
}

int g_sat_cus[64];                          // compute units per device (0 = not asked yet)

// workgroups per CU by pass: the statistics pass fits 128 registers (4 waves per SIMD), the max pass 168 (3), the backward pass 256 (2)
constexpr int SAT_PER_CU[3] = {4, 3, 2};

int sat_grid(int64_t M, int mode) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int n = __atomic_load_n(&g_sat_cus[dev], __ATOMIC_RELAXED);
  if (n == 0) {
    hipDeviceProp_t p;
    n = hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    __atomic_store_n(&g_sat_cus[dev], n, __ATOMIC_RELAXED);
  }
  const int64_t want = (M + 3) / 4, cap = (int64_t)SAT_PER_CU[mode] * n;
  return (int)(want < cap ? want : cap);
}

template <int H1, int H2, int MODE>
int launch_sat(const SaTrainArgs& a, int grid, hipStream_t st) {
  constexpr int bytes = SaLds<H1, H2, MODE>::TOTAL * 4;
  if (bytes > 64 * 1024) {
    static bool done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!__atomic_load_n(&done[dev], __ATOMIC_ACQUIRE)) {
      if (hipFuncSetAttribute((const void*)sa_train_kernel<H1, H2, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) !=
          hipSuccess)
        return CRB_ERR_LAUNCH;
      __atomic_store_n(&done[dev], true, __ATOMIC_RELEASE);
    }
  }
  hipLaunchKernelGGL((sa_train_kernel<H1, H2, MODE>), dim3(grid), dim3(256), bytes, st, a);
  return CRB_OK;
}

template <int MODE>
int dispatch_sat(int h1, int h2, const SaTrainArgs& a, int grid, hipStream_t st) {
#define CRB_SAT_CASE(A, Bc) \
  if (h1 == A && h2 == Bc) return launch_sat<A, Bc, MODE>(a, grid, st);
  CRB_SAT_CASE(16, 16) CRB_SAT_CASE(16, 32) CRB_SAT_CASE(16, 64)
  CRB_SAT_CASE(32, 16) CRB_SAT_CASE(32, 32) CRB_SAT_CASE(32, 64)
  CRB_SAT_CASE(64, 16) CRB_SAT_CASE(64, 32) CRB_SAT_CASE(64, 64)
#undef CRB_SAT_CASE
  return CRB_ERR_UNSUPPORTED;
}

bool sat_common_ok(int B, int64_t M, int ns, int h1, int h2) {
  return B > 0 && M > 0 && ns > 0 && crb_sa_mlp2_train_supported(h1, h2, ns) && M * ns < (1LL << 31);
}

}  // namespace

extern "C" int crb_sa_mlp2_train_supported(int h1, int h2, int nsample) {
  return (h1 == 16 || h1 == 32 || h1 == 64) && (h2 == 16 || h2 == 32 || h2 == 64) && nsample >= 16 && nsample % 16 == 0;
}

extern "C" int64_t crb_sa_mlp2_train_waves(int64_t M) { return (int64_t)sat_grid(M < 1 ? 1 : M, 0) * 4; }

extern "C" int crb_sa_mlp2_train_stats(int B, int64_t M, int nsample, int h1, int h2, const float* xyz,
                                       const int32_t* xyz_batch_cnt, const float* P, const float* new_xyz,
                                       const int32_t* new_xyz_batch_cnt, const int32_t* idx, const uint8_t* empty_mask,
                                       const float* W1x, const float* mean1, const float* invstd1, const float* gamma1,
                                       const float* beta1, const float* W2, float* wave_sums, void* stream) {
  if (!sat_common_ok(B, M, nsample, h1, h2) || !wave_sums) return CRB_ERR_ARG;
  SaTrainArgs a{};
  a.skip = g_sat_skip;
  a.B = B; a.M = M; a.ns = nsample; a.xyz = xyz; a.xyz_cnt = xyz_batch_cnt; a.P = P; a.new_xyz = new_xyz;
  a.new_cnt = new_xyz_batch_cnt; a.idx = idx; a.empty = empty_mask; a.W1x = W1x; a.W2 = W2;
  a.bn1[0] = mean1; a.bn1[1] = invstd1; a.bn1[2] = gamma1; a.bn1[3] = beta1;
  a.stat = wave_sums;
  const int rc = dispatch_sat<0>(h1, h2, a, sat_grid(M, 0), (hipStream_t)stream);
  if (rc != CRB_OK) return rc;
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_sa_mlp2_train_max(int B, int64_t M, int nsample, int h1, int h2, const float* xyz,
                                     const int32_t* xyz_batch_cnt, const float* P, const float* new_xyz,
                                     const int32_t* new_xyz_batch_cnt, const int32_t* idx, const uint8_t* empty_mask,
                                     const float* W1x, const float* mean1, const float* invstd1, const float* gamma1,
                                     const float* beta1, const float* W2, const float* mean2, const float* invstd2,
                                     const float* gamma2, const float* beta2, float* out, int64_t out_row_stride, int32_t* arg,
                                     float* y_sel, void* stream) {
  if (!sat_common_ok(B, M, nsample, h1, h2) || !out || !arg || !y_sel || out_row_stride < h2) return CRB_ERR_ARG;
  SaTrainArgs a{};
  a.skip = g_sat_skip;
  a.B = B; a.M = M; a.ns = nsample; a.xyz = xyz; a.xyz_cnt = xyz_batch_cnt; a.P = P; a.new_xyz = new_xyz;
  a.new_cnt = new_xyz_batch_cnt; a.idx = idx; a.empty = empty_mask; a.W1x = W1x; a.W2 = W2;
  a.bn1[0] = mean1; a.bn1[1] = invstd1; a.bn1[2] = gamma1; a.bn1[3] = beta1;
  a.bn2[0] = mean2; a.bn2[1] = invstd2; a.bn2[2] = gamma2; a.bn2[3] = beta2;
  a.out = out; a.ld_out = out_row_stride; a.arg = arg; a.ysel = y_sel;
  const int rc = dispatch_sat<1>(h1, h2, a, sat_grid(M, 1), (hipStream_t)stream);
  if (rc != CRB_OK) return rc;
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int64_t crb_sa_mlp2_train_backward_workspace_floats(int64_t M, int h1, int h2) {
  const int64_t grid = sat_grid(M < 1 ? 1 : M, 2);
  return grid * 4 * 2 * h1 + grid * (int64_t)h1 * h2;
}

extern "C" int crb_sa_mlp2_train_backward(int B, int64_t M, int nsample, int h1, int h2, const float* xyz,
                                          const int32_t* xyz_batch_cnt, const float* P, const float* new_xyz,
                                          const int32_t* new_xyz_batch_cnt, const int32_t* idx, const uint8_t* empty_mask,
                                          const float* W1x, const float* mean1, const float* invstd1, const float* gamma1,
                                          const float* beta1, const float* W2, const float* mean2, const float* invstd2,
                                          const float* gamma2, const float* beta2, const float* grad_out,
                                          int64_t grad_row_stride, const int32_t* arg, const float* dbeta2, const float* dgamma2,
                                          float* grad_z1_masked, float* dsums1, float* dW2, float* workspace,
                                          int64_t workspace_floats, void* stream) {
  if (!sat_common_ok(B, M, nsample, h1, h2) || !grad_out || !arg || !grad_z1_masked || !dsums1 || !dW2 ||
      grad_row_stride < h2 || (grad_row_stride & 3))
    return CRB_ERR_ARG;
  if (!workspace || workspace_floats < crb_sa_mlp2_train_backward_workspace_floats(M, h1, h2)) return CRB_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int grid = sat_grid(M, 2);
  SaTrainArgs a{};
  a.skip = g_sat_skip;
  a.B = B; a.M = M; a.ns = nsample; a.xyz = xyz; a.xyz_cnt = xyz_batch_cnt; a.P = P; a.new_xyz = new_xyz;
  a.new_cnt = new_xyz_batch_cnt; a.idx = idx; a.empty = empty_mask; a.W1x = W1x; a.W2 = W2;
  a.bn1[0] = mean1; a.bn1[1] = invstd1; a.bn1[2] = gamma1; a.bn1[3] = beta1;
  a.bn2[0] = mean2; a.bn2[1] = invstd2; a.bn2[2] = gamma2; a.bn2[3] = beta2;
  a.gout = grad_out; a.ld_g = grad_row_stride; a.arg = const_cast<int32_t*>(arg); a.dbeta2 = dbeta2; a.dgamma2 = dgamma2;
  a.inv_n = 1.0f / (float)(M * nsample);
  a.gz1 = grad_z1_masked;
  a.part1 = workspace;
  a.partW = workspace + (int64_t)grid * 4 * 2 * h1;
  const int rc = dispatch_sat<2>(h1, h2, a, grid, st);
  if (rc != CRB_OK) return rc;
  const SaReduceJob ja{a.part1, (int64_t)grid * 4, 2 * h1, dsums1}, jb{a.partW, (int64_t)grid, h1 * h2, dW2};
  const int blocksA = crb_cdiv(2 * h1, 16), blocksB = crb_cdiv(h1 * h2, 16);
  hipLaunchKernelGGL(sa_reduce_parts_kernel, dim3(blocksA + blocksB), dim3(256), 0, st, ja, blocksA, jb);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

#ifdef CRB_MEASURE
extern "C" int crb_sa_mlp2_train_set_skip(int bits) {
  g_sat_skip = bits;
  return CRB_OK;
}
#endif
