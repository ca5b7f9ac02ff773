// Inference-time set abstraction for gfx950: grouping -> shared 2-layer MLP -> max over the neighbourhood in ONE kernel.
//
// Replaces, for eval / no-grad calls, the body of StackSAModuleMSG.forward
// (pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py:73-112): QueryAndGroup's (M, 3+C, ns) tensor
// (pointnet2_utils.py:107-155), the two Conv2d(1x1)+BatchNorm2d+ReLU layers and F.max_pool2d over ns. At the RoI-grid
// pooling shape of PV-RCNN (M = 16 x 128 x 216 queries, ns = 16, C = 128) the grouped tensor alone is 3.7 GB per radius
// and the module path makes ~8 passes over tensors of that size; the MLP output before the max is never needed.
//
// Algebra used (the host folds BN into the convs, w' = w*gamma/sqrt(var+eps), b' = beta - mean*that):
//   layer 1 is linear before its ReLU, so   W1 [xyz_j - c_i ; f_j] + b1  =  W1x (xyz_j - c_i)  +  (W1f f_j)  +  b1
//   with P = F W1f^T computed ONCE per source point (N x H1, a small GEMM done by the caller) instead of once per
//   (query, sample) pair. xyz_j - c_i is formed exactly like the reference forms it (no large-coordinate cancellation).
//   layer 2 runs on the f32 MFMA: one wave owns one query, its 16 samples are the 16 rows of a 16x16x4 tile, W2 lives in
//   registers for the whole kernel, the max over samples is a 4-register + 2-shuffle reduction of the accumulator.
//   Empty balls: the reference zeroes the grouped features, so every row is relu(b1) -> one constant vector per launch.
//
// Numerics: same products as the module path, summed in a different order (f32): compared in tests at rtol 1e-4.
#include "crb_common.h"

namespace {

__device__ __forceinline__ int sa_locate(const int* __restrict__ cnt, int B, int64_t i, const int* __restrict__ other_cnt) {
  int acc = cnt[0], os = 0;
  for (int k = 1; k < B; ++k) {
    if (i < acc) break;
    acc += cnt[k];
    os += other_cnt[k - 1];
  }
  return os;
}

typedef float f4 __attribute__((ext_vector_type(4)));

template <int H1, int H2>
__global__ __launch_bounds__(256) void sa_mlp2_max_kernel(
    int B, int64_t M, int ns, const float* __restrict__ xyz, const int* __restrict__ xyz_cnt,
    const float* __restrict__ P /* (N,H1) */, const float* __restrict__ new_xyz, const int* __restrict__ new_cnt,
    const int* __restrict__ idx, const unsigned char* __restrict__ empty, const float* __restrict__ W1x /* (3,H1) */,
    const float* __restrict__ b1, const float* __restrict__ W2 /* (H1,H2) */, const float* __restrict__ b2,
    float* __restrict__ out, int out_stride) {
  constexpr int KPL = H1 / 4;       // k values per lane group: lane (r,g) owns k = KPL*g .. KPL*g+KPL-1
  constexpr int NB = H2 / 16;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int T = (ns + 15) >> 4;     // row tiles per query

  float w2[KPL][NB], w1x[3][KPL], bb1[KPL], bb2[NB];
#pragma unroll
  for (int s = 0; s < KPL; ++s) {
    const int k = KPL * g + s;
    bb1[s] = b1[k];
#pragma unroll
    for (int d = 0; d < 3; ++d) w1x[d][s] = W1x[d * H1 + k];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) w2[s][nb] = W2[k * H2 + 16 * nb + r];
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) bb2[nb] = b2[16 * nb + r];

  // all 16 rows equal relu(b1): the max over rows is that row's product
  float cst[NB];
  {
    f4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KPL; ++s) {
      const float h = fmaxf(bb1[s], 0.f);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(h, w2[s][nb], acc[nb], 0, 0, 0);
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) cst[nb] = fmaxf(acc[nb][0] + bb2[nb], 0.f);
  }

  const int64_t nw = (int64_t)gridDim.x * 4;
  const int64_t w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t ntile = M * T;

  // stage A (two tiles ahead): source row + query centre; stage B (one tile ahead): P row slice + offset vector
  struct StA { int row; float cx, cy, cz; };
  struct StB { f4 p[KPL / 4]; float dx, dy, dz; int row; };
  auto load_a = [&](int64_t tile) {
    StA a;
    a.row = -1; a.cx = a.cy = a.cz = 0.f;
    if (tile < ntile) {
      const int64_t q = tile / T;
      const int t = (int)(tile - q * T);
      if (!empty[q]) {
        int smp = 16 * t + r;
        smp = smp < ns ? smp : ns - 1;                        // padding rows repeat a real sample: the max is unchanged
        a.row = sa_locate(new_cnt, B, q, xyz_cnt) + idx[q * ns + smp];
      }
      a.cx = new_xyz[q * 3 + 0]; a.cy = new_xyz[q * 3 + 1]; a.cz = new_xyz[q * 3 + 2];
    }
    return a;
  };
  auto load_b = [&](const StA& a) {
    StB b;
    b.row = a.row;
    b.dx = b.dy = b.dz = 0.f;
#pragma unroll
    for (int v = 0; v < KPL / 4; ++v) b.p[v] = f4{0.f, 0.f, 0.f, 0.f};
    if (a.row >= 0) {
      const f4* src = reinterpret_cast<const f4*>(P + (int64_t)a.row * H1 + KPL * g);
#pragma unroll
      for (int v = 0; v < KPL / 4; ++v) b.p[v] = src[v];
      b.dx = xyz[(int64_t)a.row * 3 + 0] - a.cx;
      b.dy = xyz[(int64_t)a.row * 3 + 1] - a.cy;
      b.dz = xyz[(int64_t)a.row * 3 + 2] - a.cz;
    }
    return b;
  };

  // this wave's i-th tile: query w0 + (i/T)*nw, row tile i%T (tiles of one query stay on one wave, in order)
  auto tile_of = [&](int64_t i) {
    const int64_t q = w0 + (i / T) * nw;
    return q < M ? q * T + (i % T) : ntile;
  };
  StA a_nxt = load_a(tile_of(0));
  StB b_cur = load_b(a_nxt);
  a_nxt = load_a(tile_of(1));
  float run[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) run[nb] = -3.4e38f;

  int64_t i = 0;
  for (int64_t q = w0; q < M; q += nw) {
    for (int t = 0; t < T; ++t, ++i) {
      StB b_nxt = load_b(a_nxt);                              // tile i+1
      a_nxt = load_a(tile_of(i + 2));
      const bool live = __builtin_amdgcn_readfirstlane(b_cur.row) >= 0;   // empty[q] is per query: wave-uniform
      if (live) {
        f4 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KPL; ++s) {
          float h = b_cur.p[s >> 2][s & 3] + bb1[s];
          h = fmaf(w1x[0][s], b_cur.dx, h);
          h = fmaf(w1x[1][s], b_cur.dy, h);
          h = fmaf(w1x[2][s], b_cur.dz, h);
          h = fmaxf(h, 0.f);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(h, w2[s][nb], acc[nb], 0, 0, 0);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          float m = fmaxf(fmaxf(acc[nb][0], acc[nb][1]), fmaxf(acc[nb][2], acc[nb][3]));
          m = fmaxf(m, __shfl_xor(m, 16));
          m = fmaxf(m, __shfl_xor(m, 32));
          run[nb] = fmaxf(run[nb], m);
        }
      }
      if (t == T - 1) {
        if (g == 0) {
          float* o = out + q * out_stride + r;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) o[16 * nb] = live ? fmaxf(run[nb] + bb2[nb], 0.f) : cst[nb];
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) run[nb] = -3.4e38f;
      }
      b_cur = b_nxt;
    }
  }
}

template <int H1, int H2>
void launch_sa(int B, int64_t M, int ns, const float* xyz, const int* xyz_cnt, const float* P, const float* new_xyz,
               const int* new_cnt, const int* idx, const unsigned char* empty, const float* W1x, const float* b1,
               const float* W2, const float* b2, float* out, int out_stride, hipStream_t st) {
  const int64_t want = (M + 3) / 4;
  const int grid = (int)(want < 4096 ? want : 4096);
  hipLaunchKernelGGL((sa_mlp2_max_kernel<H1, H2>), dim3(grid), dim3(256), 0, st, B, M, ns, xyz, xyz_cnt, P, new_xyz,
                     new_cnt, idx, empty, W1x, b1, W2, b2, out, out_stride);
}

}  // namespace

extern "C" int crb_sa_mlp2_max_supported(int h1, int h2) {
  return (h1 == 16 || h1 == 32 || h1 == 64) && (h2 == 16 || h2 == 32 || h2 == 64);
}

extern "C" int crb_sa_mlp2_max_stack(int B, int64_t M, int nsample, int h1, int h2, const float* xyz,
                                     const int32_t* xyz_batch_cnt, const float* P, const float* new_xyz,
                                     const int32_t* new_xyz_batch_cnt, const int32_t* idx, const uint8_t* empty_mask,
                                     const float* W1x, const float* b1, const float* W2, const float* b2, float* out,
                                     int out_stride, void* stream) {
  if (B <= 0 || M < 0 || nsample <= 0 || out_stride < h2) return CRB_ERR_ARG;
  if (!crb_sa_mlp2_max_supported(h1, h2)) return CRB_ERR_UNSUPPORTED;
  if (M == 0) return CRB_OK;
  hipStream_t st = (hipStream_t)stream;
#define CRB_SA_CASE(A, Bc)                                                                                          \
  if (h1 == A && h2 == Bc) {                                                                                        \
    launch_sa<A, Bc>(B, M, nsample, xyz, xyz_batch_cnt, P, new_xyz, new_xyz_batch_cnt, idx, empty_mask, W1x, b1, W2, \
                     b2, out, out_stride, st);                                                                      \
  }
  CRB_SA_CASE(16, 16) CRB_SA_CASE(16, 32) CRB_SA_CASE(16, 64)
  CRB_SA_CASE(32, 16) CRB_SA_CASE(32, 32) CRB_SA_CASE(32, 64)
  CRB_SA_CASE(64, 16) CRB_SA_CASE(64, 32) CRB_SA_CASE(64, 64)
#undef CRB_SA_CASE
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}
