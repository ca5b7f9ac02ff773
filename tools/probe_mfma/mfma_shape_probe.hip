// Probe (not part of the product): what fraction of the f32 matrix pipe does the BARE inner loop of the Winograd kernels reach -
// operand reads from LDS + MFMAs + one barrier per chunk, no transform, no DMA - with v_mfma_f32_16x16x4_f32 (the kernels' tiling:
// 64 MFMAs + 24 ds_read_b128 per wave and chunk) against v_mfma_f32_32x32x2_f32 (32 MFMAs + 16 ds_read_b128 for the same flops)?
// 512 threads = two waves per SIMD, 128 accumulator registers per wave in both. Build + run: see run.sh.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, bool BARRIER>
__global__ __launch_bounds__(512, 2) void probe(float* out, int chunks, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int T = threadIdx.x, lane = T & 63;
  for (int i = T; i < 16384; i += 512) lds[i] = 1e-3f * (float)((i * 37) & 255);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float total = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 acc[16][2];
#pragma unroll
    for (int x = 0; x < 16; ++x) acc[x][0] = acc[x][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* base = lds + lane * 4;
    for (int g = 0; g < chunks; ++g) {
      const float* U = base + (g & 1) * 8192;
      f32x4 ua[2], v0[2], v1[2];
      ua[0] = *(const f32x4*)(U); v0[0] = *(const f32x4*)(U + 256); v1[0] = *(const f32x4*)(U + 512);
#pragma unroll
      for (int xp = 0; xp < 8; ++xp) {
        const int s = xp & 1, n = s ^ 1;
        if (xp < 7) {
          ua[n] = *(const f32x4*)(U + (xp + 1) * 1024);
          v0[n] = *(const f32x4*)(U + (xp + 1) * 1024 + 256);
          v1[n] = *(const f32x4*)(U + (xp + 1) * 1024 + 512);
        }
        if (BARRIER && xp == 7) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int xi = 2 * xp + h;
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[s][2 * h + e], v0[s][2 * h + e], acc[xi][0], 0, 0, 0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[s][2 * h + e], v1[s][2 * h + e], acc[xi][1], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int x = 0; x < 16; ++x) total += acc[x][0][0] + acc[x][1][3];
  } else {
    f32x16 acc[8];
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    const float* base = lds + lane * 4;
    for (int g = 0; g < chunks; ++g) {
      const float* U = base + (g & 1) * 8192;
      f32x4 a[2][2], b[2][2];        // [slot][xi of the pair]
      a[0][0] = *(const f32x4*)(U); b[0][0] = *(const f32x4*)(U + 256); a[0][1] = *(const f32x4*)(U + 512); b[0][1] = *(const f32x4*)(U + 768);
#pragma unroll
      for (int xp = 0; xp < 4; ++xp) {      // xi pairs: 4 stages of 8 MFMAs (64 cycles each)
        const int s = xp & 1, n = s ^ 1;
        if (xp < 3) {
          a[n][0] = *(const f32x4*)(U + (xp + 1) * 2048); b[n][0] = *(const f32x4*)(U + (xp + 1) * 2048 + 256);
          a[n][1] = *(const f32x4*)(U + (xp + 1) * 2048 + 512); b[n][1] = *(const f32x4*)(U + (xp + 1) * 2048 + 768);
        }
        if (BARRIER && xp == 3) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            acc[2 * xp + h] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][h][k], b[s][h][k], acc[2 * xp + h], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int x = 0; x < 8; ++x) total += acc[x][0] + acc[x][15];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[(size_t)blockIdx.x * 512 + T] = total;
  if (T == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE, bool BARRIER>
static void run(const char* name, float* out, unsigned long long* cyc, int chunks) {
  hipFuncSetAttribute((const void*)probe<SHAPE, BARRIER>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<SHAPE, BARRIER>), dim3(256), dim3(512), 160 * 1024, 0, out, chunks, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0;
  for (int i = 0; i < 256; ++i) mean += (double)h[i];
  mean /= 256.0;
  // s_memtime counts at a fixed 100 MHz on this chip: use the event time and the MFMA work instead of cycles
  const double flops = 256.0 * 8 * chunks * 64 * 2048.0;            // workgroups x waves x chunks x (64 MFMAs of 2048 flop | 32 of 4096)
  printf("%-44s %8.1f us  %6.1f TF = %.3f of the 157.3 TF f32 MFMA peak\n", name, ms * 1e3, flops / (ms * 1e-3) / 1e12,
         flops / (ms * 1e-3) / 1e12 / 157.3);
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  const int chunks = 4096;
  run<16, true>("16x16x4, 24 b128 reads, barrier per chunk", out, cyc, chunks);
  run<16, false>("16x16x4, 24 b128 reads, no barrier", out, cyc, chunks);
  run<32, true>("32x32x2, 16 b128 reads, barrier per chunk", out, cyc, chunks);
  run<32, false>("32x32x2, 16 b128 reads, no barrier", out, cyc, chunks);
  return 0;
}
