// Latency-floor probes for the low-channel subm gather-GEMM (MEASUREMENT library only, tools/lowchannel_floor2.py).
// The level-1 layers (16 channels, 227 k rows, 3.5 neighbours per row) move 35 MB in 17 us: what bounds them is not bandwidth but the
// number of DEPENDENT global round trips between "the row index is known" and "the output row is stored". These kernels do the
// memory side of that chain and nothing else (no weights, no MFMA, no table decoding beyond what the chain needs):
//   variant 0  copy              y[i] = x[i]                                             1 round trip
//   variant 1  two round trips   fixed-stride neighbour list ell[i][0..7] -> rows x[ell] (summed) -> y[i]
//   variant 2  three round trips cmask[i], cbase[i] -> packed[cbase[i] ..] -> rows (summed) -> y[i]   (the compact table's chain)
//   variant 3  input-stationary  x[i] read ONCE, fixed-stride neighbour list ell[i][0..7] -> atomicAdd of the row into y[ell] (round 6,
//              the formulation that moves the "algorithmic" bytes only - what does its scatter cost?); y pre-zeroed
//   variant 4  the same with the rows of one workgroup (64 rows, consecutive in (b, z, y, x) order) first added into an LDS tile of
//              the workgroup's own output rows when the neighbour falls into it (its x-neighbours), global atomics for the rest
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
  pf4 acc = (pf4){0.f, 0.f, 0.f, 0.f};
  if (VARIANT == 3 || VARIANT == 4) {
    const bool live = i < n;
    __shared__ float tile[64 * 16];
    const int row0 = blockIdx.x * 64;
    if (VARIANT == 4) {
      for (int e = threadIdx.x; e < 64 * 16; e += 256) tile[e] = 0.f;
      __syncthreads();
    }
    const int ii = live ? i : 0;
    const pf4 v = *reinterpret_cast<const pf4*>(x + (int64_t)ii * 16 + 4 * q);
    const int4 e0 = *reinterpret_cast<const int4*>(ell + (int64_t)ii * 8), e1 = *reinterpret_cast<const int4*>(ell + (int64_t)ii * 8 + 4);
    const int e[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (e[k] < 0 || !live) continue;
      if (VARIANT == 4 && e[k] >= row0 && e[k] < row0 + 64) {
        float* t4 = tile + (e[k] - row0) * 16 + 4 * q;
        atomicAdd(t4 + 0, v[0]); atomicAdd(t4 + 1, v[1]); atomicAdd(t4 + 2, v[2]); atomicAdd(t4 + 3, v[3]);
      } else {
        float* o = y + (int64_t)e[k] * 16 + 4 * q;
        atomicAdd(o + 0, v[0]); atomicAdd(o + 1, v[1]); atomicAdd(o + 2, v[2]); atomicAdd(o + 3, v[3]);
      }
    }
    if (VARIANT == 4) {
      __syncthreads();
      if (!live) return;
      float* o = y + (int64_t)i * 16 + 4 * q;
      const float* t4 = tile + (i - row0) * 16 + 4 * q;
      atomicAdd(o + 0, t4[0]); atomicAdd(o + 1, t4[1]); atomicAdd(o + 2, t4[2]); atomicAdd(o + 3, t4[3]);
    }
    return;
  }
  if (i >= n) return;
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
// ---- calibration of rocprofv3's FETCH_SIZE on the Winograd forward kernel's access pattern: LDS-DMA reads of
// KNOWN bytes. Every lane issues global_load_lds_dwordx4 (16 bytes); a wave instruction covers 32 pieces of 32 bytes (two lanes per
// piece), the pieces `stride` bytes apart - stride 32: a dense streaming read; stride 512: the 8-channel pieces of adjacent pixels
// of a 128-channel NHWC map (what the raw-block DMA of winograd2_kernel reads). pass_mask bit p: also read the p-th 32-byte piece of
// every line in a later sweep (bits 0..3: all four 8-channel chunks of a 128-byte line, as the kernel's chunk loop does over time).
// Every requested byte is requested exactly once per set bit. The landed data is summed into one float per workgroup (kept alive).
__global__ __launch_bounds__(256) void probe_lds_dma_kernel(const float* __restrict__ src, int64_t pieces, int stride_floats,
                                                             int pass_mask, float* __restrict__ sink) {
  __shared__ __attribute__((aligned(16))) float buf[4][256];            // 1 KiB per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t nwave = (int64_t)gridDim.x * 4, w = (int64_t)blockIdx.x * 4 + wave;
  float acc = 0.f;
  for (int p = 0; p < 4; ++p) {
    if (!((pass_mask >> p) & 1)) continue;
    for (int64_t g = w; g * 32 < pieces; g += nwave) {                   // 32 pieces per wave instruction
      const int64_t piece = g * 32 + (lane >> 1);
      const float* a = src + (piece < pieces ? piece : 0) * stride_floats + p * 8 + (lane & 1) * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a,
                                       (__attribute__((address_space(3))) void*)buf[wave], 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc += buf[wave][lane * 4];
    }
  }
  acc = crb_wave_sum(acc);
  if (lane == 0) atomicAdd(&sink[blockIdx.x & 255], acc);
}

// ---- how fast can ONE workgroup per CU stream an L2-resident image into LDS? (round 6: the split-bf16 Winograd kernel asks for 30 KB
// per ~1,000 cycles and CU.) Every workgroup walks the same `bytes`-long image `iters` times, 16 bytes per lane and instruction,
// DEPTH instructions in flight per wave. VARIANT 0: LDS-DMA (global_load_lds_dwordx4); 1: global_load_dwordx4 into registers, then
// ds_write_b128 (register staging). The landed data is never read (the write into LDS is what is timed).
template <int VARIANT, int DEPTH>
__global__ __launch_bounds__(512) void probe_stream_kernel(const float* __restrict__ src, int64_t bytes, int iters, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) float sbuf[];            // DEPTH KiB per wave
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  float* my = sbuf + wave * DEPTH * 256;
  const int64_t per_sweep = bytes / (1024 * (int64_t)nw * DEPTH);         // groups of DEPTH KiB per wave and sweep
  pf4 keep = (pf4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    for (int64_t g = 0; g < per_sweep; ++g) {
      const float* base = src + ((g * nw + wave) * DEPTH) * 256 + lane * 4;
      if (VARIANT == 0) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + d * 256),
                                           (__attribute__((address_space(3))) void*)(my + d * 256), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH / 2) : "memory");      // half of them stay in flight across the loop edge
      } else {
        pf4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = *reinterpret_cast<const pf4*>(base + d * 256);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) *reinterpret_cast<pf4*>(my + d * 256 + lane * 4) = v[d];
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  keep = *reinterpret_cast<const pf4*>(my + lane * 4);
  if (keep[0] == 123.456f) sink[0] = keep[1];
}
}  // namespace

// variant 0 LDS-DMA / 1 register staging; threads 256 or 512 per workgroup, one workgroup per CU (grid = cus); depth 4 or 8
extern "C" int crb_probe_stream(int variant, int threads, int depth, int cus, const float* src, int64_t bytes, int iters, float* sink,
                                void* stream) {
  if (!src || !sink || bytes <= 0 || iters <= 0 || cus <= 0 || (threads != 256 && threads != 512) || (depth != 4 && depth != 8)) return CRB_ERR_ARG;
  const size_t lds = (size_t)(threads / 64) * depth * 1024;
  hipStream_t st = (hipStream_t)stream;
#define CRB_PS(V, D) hipLaunchKernelGGL((probe_stream_kernel<V, D>), dim3(cus), dim3(threads), lds, st, src, bytes, iters, sink)
  if (variant == 0 && depth == 4) CRB_PS(0, 4);
  else if (variant == 0) CRB_PS(0, 8);
  else if (depth == 4) CRB_PS(1, 4);
  else CRB_PS(1, 8);
#undef CRB_PS
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_probe_lds_dma(const float* src, int64_t pieces, int stride_bytes, int pass_mask, float* sink256, void* stream) {
  if (!src || !sink256 || pieces <= 0 || stride_bytes < 32 || (stride_bytes & 15) || !(pass_mask & 15)) return CRB_ERR_ARG;
  hipLaunchKernelGGL(probe_lds_dma_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, src, pieces, stride_bytes / 4, pass_mask,
                     sink256);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_probe_gather_chain(int variant, const float* x, int64_t n, const uint32_t* cmask, const int32_t* cbase,
                                      const int32_t* packed, const int32_t* ell, float* y, void* stream) {
  if (n <= 0 || n >= (1 << 29) || !x || !y) return CRB_ERR_ARG;
  const dim3 grid(crb_cdiv(n * 4, 256));
  hipStream_t st = (hipStream_t)stream;
  if (variant == 0) hipLaunchKernelGGL(probe_chain_kernel<0>, grid, dim3(256), 0, st, x, (int)n, cmask, cbase, packed, ell, y);
  else if (variant == 1 && ell) hipLaunchKernelGGL(probe_chain_kernel<1>, grid, dim3(256), 0, st, x, (int)n, cmask, cbase, packed, ell, y);
  else if (variant == 2 && cmask && cbase && packed) hipLaunchKernelGGL(probe_chain_kernel<2>, grid, dim3(256), 0, st, x, (int)n, cmask, cbase, packed, ell, y);
  else if (variant == 3 && ell) hipLaunchKernelGGL(probe_chain_kernel<3>, grid, dim3(256), 0, st, x, (int)n, cmask, cbase, packed, ell, y);
  else if (variant == 4 && ell) hipLaunchKernelGGL(probe_chain_kernel<4>, grid, dim3(256), 0, st, x, (int)n, cmask, cbase, packed, ell, y);
  else return CRB_ERR_ARG;
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}
#endif  // CRB_MEASURE
