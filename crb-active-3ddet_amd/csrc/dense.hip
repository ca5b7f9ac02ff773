// SparseConvTensor.dense() / HeightCompression scatter and its backward gather (row a6 of SURVEY §8).
// Replaces spconv's SparseConvTensor.dense() as used by
// pcdet/models/backbones_2d/map_to_bev/height_compression.py:20-24.
//   out (B, C, D, H, W) f32, zero filled then out[b,c,z,y,x] = feat[row,c]
// One 64-row x 64-channel tile per workgroup, staged through LDS so that HBM reads are row-contiguous
// and writes run along x (rows are in ascending (b,z,y,x) order after a strided conv, so consecutive rows
// are mostly consecutive x).
#include "crb_common.h"
#include "../../include/crb_hip.h"

namespace {

__global__ __launch_bounds__(256) void dense_scatter_kernel(const float* __restrict__ feat, const int* __restrict__ coords,
                                                            float* __restrict__ out, int n, int C, int D, int H, int W) {
  __shared__ float tile[64][65];
  __shared__ int64_t base[64];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  for (int t = threadIdx.x; t < 64 * 64; t += 256) {
    int r = t >> 6, c = t & 63;
    float v = 0.f;
    if (r0 + r < n && c0 + c < C) v = feat[(int64_t)(r0 + r) * C + c0 + c];
    tile[r][c] = v;
  }
  if (threadIdx.x < 64) {
    int r = r0 + threadIdx.x;
    int64_t b = -1;
    if (r < n) {
      int4 q = *reinterpret_cast<const int4*>(coords + (int64_t)r * 4);
      b = (((int64_t)q.x * C) * D + q.y) * H * (int64_t)W + (int64_t)q.z * W + q.w;   // channel 0 address
    }
    base[threadIdx.x] = b;
  }
  __syncthreads();
  const int64_t cstride = (int64_t)D * H * W;
  for (int t = threadIdx.x; t < 64 * 64; t += 256) {
    int c = t >> 6, r = t & 63;      // lanes run over rows -> consecutive x
    int64_t b = base[r];
    if (b >= 0 && c0 + c < C) out[b + (int64_t)(c0 + c) * cstride] = tile[r][c];
  }
}

__global__ __launch_bounds__(256) void dense_gather_kernel(const float* __restrict__ dense, const int* __restrict__ coords,
                                                           float* __restrict__ feat, int n, int C, int D, int H, int W) {
  __shared__ float tile[64][65];
  __shared__ int64_t base[64];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  if (threadIdx.x < 64) {
    int r = r0 + threadIdx.x;
    int64_t b = -1;
    if (r < n) {
      int4 q = *reinterpret_cast<const int4*>(coords + (int64_t)r * 4);
      b = (((int64_t)q.x * C) * D + q.y) * H * (int64_t)W + (int64_t)q.z * W + q.w;
    }
    base[threadIdx.x] = b;
  }
  __syncthreads();
  const int64_t cstride = (int64_t)D * H * W;
  for (int t = threadIdx.x; t < 64 * 64; t += 256) {
    int c = t >> 6, r = t & 63;
    int64_t b = base[r];
    tile[r][c] = (b >= 0 && c0 + c < C) ? dense[b + (int64_t)(c0 + c) * cstride] : 0.f;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 64 * 64; t += 256) {
    int r = t >> 6, c = t & 63;
    if (r0 + r < n && c0 + c < C) feat[(int64_t)(r0 + r) * C + c0 + c] = tile[r][c];
  }
}

// channels-last BEV layout (B, H, W, C*D), channel index c*D + z: what MIOpen's NHWC igemm kernels consume without the
// NCHW<->NHWC transposes it otherwise inserts around them. One thread per (row, c): reads run along c, the C values of a
// row land in one (C*D)-float span.
__global__ __launch_bounds__(256) void dense_scatter_nhwc_kernel(const float* __restrict__ feat, const int* __restrict__ coords,
                                                                 float* __restrict__ out, int64_t total, int C, int D, int H,
                                                                 int W) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int64_t r = t / C;
  const int c = (int)(t - r * C);
  const int4 q = *reinterpret_cast<const int4*>(coords + r * 4);
  out[((((int64_t)q.x * H + q.z) * W + q.w) * C + c) * D + q.y] = feat[t];
}

__global__ __launch_bounds__(256) void dense_gather_nhwc_kernel(const float* __restrict__ dense, const int* __restrict__ coords,
                                                                float* __restrict__ feat, int64_t total, int C, int D, int H,
                                                                int W) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int64_t r = t / C;
  const int c = (int)(t - r * C);
  const int4 q = *reinterpret_cast<const int4*>(coords + r * 4);
  feat[t] = dense[((((int64_t)q.x * H + q.z) * W + q.w) * C + c) * D + q.y];
}

}  // namespace

extern "C" int crb_sparse_to_dense_nhwc(const float* feat, const int32_t* coords, float* out, int64_t n, int B, int C,
                                        int D, int H, int W, int zero_fill, void* stream) {
  if (n < 0 || B <= 0 || C <= 0) return CRB_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (zero_fill) CRB_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * C * D * H * W, st));
  if (n == 0) return CRB_OK;
  hipLaunchKernelGGL(dense_scatter_nhwc_kernel, dim3(crb_cdiv(n * C, 256)), dim3(256), 0, st, feat, coords, out, n * C, C,
                     D, H, W);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_dense_to_sparse_nhwc(const float* dense, const int32_t* coords, float* feat, int64_t n, int B, int C,
                                        int D, int H, int W, void* stream) {
  if (n < 0 || B <= 0 || C <= 0) return CRB_ERR_ARG;
  if (n == 0) return CRB_OK;
  hipLaunchKernelGGL(dense_gather_nhwc_kernel, dim3(crb_cdiv(n * C, 256)), dim3(256), 0, (hipStream_t)stream, dense,
                     coords, feat, n * C, C, D, H, W);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_sparse_to_dense(const float* feat, const int32_t* coords, float* out, int64_t n, int B, int C,
                                   int D, int H, int W, int zero_fill, void* stream) {
  if (n < 0 || B <= 0 || C <= 0) return CRB_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (zero_fill) CRB_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * C * D * H * W, st));
  if (n == 0) return CRB_OK;
  hipLaunchKernelGGL(dense_scatter_kernel, dim3(crb_cdiv(n, 64), crb_cdiv(C, 64)), dim3(256), 0, st, feat, coords, out,
                     (int)n, C, D, H, W);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_dense_to_sparse(const float* dense, const int32_t* coords, float* feat, int64_t n, int B, int C,
                                   int D, int H, int W, void* stream) {
  if (n < 0 || B <= 0 || C <= 0) return CRB_ERR_ARG;
  if (n == 0) return CRB_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(dense_gather_kernel, dim3(crb_cdiv(n, 64), crb_cdiv(C, 64)), dim3(256), 0, st, dense, coords, feat,
                     (int)n, C, D, H, W);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}
