// Latency-floor probes for the low-channel subm gather-GEMM (MEASUREMENT library only, tools/lowchannel_floor2.py).
// The level-1 layers (16 channels, 227 k rows, 3.5 neighbours per row) move 35 MB in 17 us: what bounds them is not bandwidth but the
// number of DEPENDENT global round trips between "the row index is known" and "the output row is stored". These kernels do the
// memory side of that chain and nothing else (no weights, no MFMA, no table decoding beyond what the chain needs):
//   variant 0  copy              y[i] = x[i]                                             1 round trip
//   variant 1  two round trips   fixed-stride neighbour list ell[i][0..7] -> rows x[ell] (summed) -> y[i]
//   variant 2  three round trips cmask[i], cbase[i] -> packed[cbase[i] ..] -> rows (summed) -> y[i]   (the compact table's chain)
// One lane per (row, channel quad): a row of 16 floats is one 64-byte segment, read / written by 4 lanes of 16 bytes.
#ifdef CRB_MEASURE
#include "crb_common.h"
#include "../../include/crb_hip.h"
#include "../../include/crb_hip_measure.h"

namespace {
typedef float pf4 __attribute__((ext_vector_type(4)));

template <int VARIANT>
__global__ __launch_bounds__(256) void probe_chain_kernel(const float* __restrict__ x, int n, const unsigned* __restrict__ cmask,
                                                          const int* __restrict__ cbase, const int* __restrict__ packed,
                                                          const int* __restrict__ ell, float* __restrict__ y) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int i = t >> 2, q = t & 3;
  if (i >= n) return;
  pf4 acc = (pf4){0.f, 0.f, 0.f, 0.f};
  if (VARIANT == 0) {
    acc = *reinterpret_cast<const pf4*>(x + (int64_t)i * 16 + 4 * q);
  } else if (VARIANT == 1) {
    const int4 e0 = *reinterpret_cast<const int4*>(ell + (int64_t)i * 8), e1 = *reinterpret_cast<const int4*>(ell + (int64_t)i * 8 + 4);
    const int e[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
    pf4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      v[k] = e[k] >= 0 ? *reinterpret_cast<const pf4*>(x + (int64_t)e[k] * 16 + 4 * q) : (pf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k];
  } else {
    const unsigned m = cmask[i];
    const int b = cbase[i], cnt = __popc(m);
    int e[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = k < cnt ? packed[b + k] : -1;
    pf4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      v[k] = e[k] >= 0 ? *reinterpret_cast<const pf4*>(x + (int64_t)e[k] * 16 + 4 * q) : (pf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k];
    for (int k = 8; k < cnt; ++k) acc += *reinterpret_cast<const pf4*>(x + (int64_t)packed[b + k] * 16 + 4 * q);
  }
  *reinterpret_cast<pf4*>(y + (int64_t)i * 16 + 4 * q) = acc;
}
}  // namespace

extern "C" int crb_probe_gather_chain(int variant, const float* x, int64_t n, const uint32_t* cmask, const int32_t* cbase,
                                      const int32_t* packed, const int32_t* ell, float* y, void* stream) {
  if (n <= 0 || n >= (1 << 29) || !x || !y) return CRB_ERR_ARG;
  const dim3 grid(crb_cdiv(n * 4, 256));
  hipStream_t st = (hipStream_t)stream;
  if (variant == 0) hipLaunchKernelGGL(probe_chain_kernel<0>, grid, dim3(256), 0, st, x, (int)n, cmask, cbase, packed, ell, y);
  else if (variant == 1 && ell) hipLaunchKernelGGL(probe_chain_kernel<1>, grid, dim3(256), 0, st, x, (int)n, cmask, cbase, packed, ell, y);
  else if (variant == 2 && cmask && cbase && packed) hipLaunchKernelGGL(probe_chain_kernel<2>, grid, dim3(256), 0, st, x, (int)n, cmask, cbase, packed, ell, y);
  else return CRB_ERR_ARG;
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}
#endif  // CRB_MEASURE
