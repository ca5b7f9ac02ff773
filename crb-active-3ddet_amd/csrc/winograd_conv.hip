// 3x3 stride-1 pad-1 convolution on channels_last (NHWC) maps as Winograd F(2x2, 3x3) on the f32 MFMA  (row a7 of SURVEY §8:
// the BEV backbone's 3x3 convolutions, pcdet/models/backbones_2d/base_bev_backbone.py:24-41, which the reference hands to
// cuDNN; MIOpen's f32 implicit-GEMM kernels run them at 130 TFLOP/s = 83 % of the exact-f32 MFMA roof, so the only way below
// 1.28 ms per 128->128 @ 200x176 x 16 call is fewer multiplications: 2.25x fewer with F(2x2,3x3)).
//
//   U[xi] = G g G^T          (16, Cin, Cout)   weights, once per call (crb_winograd_weights)
//   V[xi] = B^T d B          per 4x4 input tile and channel, in the kernel (registers -> LDS)
//   M[xi] = V[xi] U[xi]      16 independent (tiles x Cin) x (Cin x Cout) products on v_mfma_f32_32x32x2_f32
//   Y     = A^T M A          per tile and output channel, in registers on the accumulators
//
// One workgroup = 64 consecutive 2x2-output tiles (raster order over the batch) x 64 output channels; its 4 waves form a
// 2 x 2 grid (32 tiles x 32 channels each) and every wave keeps ALL 16 xi of its block — 16 accumulators of 32x32 = 256
// accumulator registers, one wave per SIMD — so the output transform needs no exchange: a lane holds the same (tile, channel)
// element of all 16 M[xi]. Cin is walked in chunks of CC = 8 channels, double-buffered in LDS (V 32 KB + U 32 KB per buffer):
// while the current chunk's MFMAs run, waves 0,1 have the next chunk's 4x4 patches in flight (a lane pair covers a tile: the
// low lane bit selects the channel quad, 16-byte loads from clamped addresses, out-of-map elements zeroed by a select) and
// waves 2,3 the next chunk's U rows; the staged values are transformed / stored after the MFMA block, then one barrier.
// (Measured variants: U straight from L2 per MFMA 4.5 ms, per-wave role branches around the MFMA block 4.8 ms (spills), every
// wave transforming patches 1.58 ms; this layout 1.125 ms per 128->128 @ 16x200x176 call — DESIGN §6.)
// Optional epilogue on the output: + bias, ReLU.
#ifdef CRB_MEASURE   // the first Winograd design (round 3): A/B reference of tools/, MEASUREMENT library only since round 5
#include "crb_common.h"
#include "../../include/crb_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int WG_TILES = 64;      // tiles per workgroup (2 wave rows x 32 = rows of the MFMA)
constexpr int WG_K = 64;          // output channels per workgroup (2 wave columns x 32)
constexpr int CC = 8;             // input channels per chunk
constexpr int V_FLOATS = 16 * CC * WG_TILES, U_FLOATS = 16 * CC * WG_K, BUF_FLOATS = V_FLOATS + U_FLOATS;

// g (3,3,Cin,Cout) -> U (16,Cin,Cout)
__global__ __launch_bounds__(256) void winograd_weights_kernel(const float* __restrict__ g, float* __restrict__ U, int cin,
                                                               int cout) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per = (int64_t)cin * cout;
  if (t >= per) return;
  float w[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) w[a][b] = g[(a * 3 + b) * per + t];
  float tmp[4][3];                 // G g
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    tmp[0][b] = w[0][b];
    tmp[1][b] = 0.5f * (w[0][b] + w[1][b] + w[2][b]);
    tmp[2][b] = 0.5f * (w[0][b] - w[1][b] + w[2][b]);
    tmp[3][b] = w[2][b];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float u0 = tmp[r][0], u1 = 0.5f * (tmp[r][0] + tmp[r][1] + tmp[r][2]),
                u2 = 0.5f * (tmp[r][0] - tmp[r][1] + tmp[r][2]), u3 = tmp[r][2];
    U[(r * 4 + 0) * per + t] = u0;
    U[(r * 4 + 1) * per + t] = u1;
    U[(r * 4 + 2) * per + t] = u2;
    U[(r * 4 + 3) * per + t] = u3;
  }
}

struct WinoArgs {
  const float* x;      // (N,H,W,Cin)
  const float* U;      // (16,Cin,Cout)
  float* y;            // (N,H,W,Cout)
  const float* bias;   // (Cout) or null
  int N, H, W, cin, cout, relu;
  int th, tw;          // tiles per column / row = ceil(H/2), ceil(W/2)
  int ntiles;          // N * th * tw
};

// LDS, double buffered per chunk of CC input channels:  V[16][CC][64 tiles] | U[16][CC][64 k]
// MODE (measurement builds, wrong results): 1 = no MFMAs, 2 = no staging of the next chunk (loads, transform, LDS stores)
template <int MODE>
__global__ __launch_bounds__(256, 1) void winograd_f2x2_3x3_kernel(WinoArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  const int wt = wave >> 1, wk = wave & 1;                           // wave's 32-tile row block / 32-channel column block
  const int tile0 = blockIdx.x * WG_TILES;
  const int k0 = blockIdx.y * WG_K;

  // ---- staging of the NEXT chunk. Patch: threads 0..127 = (tile, channel quad) with the QUAD on the lane's low bit: two
  //      neighbouring lanes read the two 16-byte halves of one pixel's 32-byte channel slice, so a wave load touches 32 lines,
  //      not 64 (the texture-address cost of these loads, not their bytes, bounded the first versions). U: waves 2,3, 16 float4 of
  //      the chunk's 16 x CC x 64 block per thread. Loads are issued at the start of a chunk and land under its MFMAs; transform and
  //      LDS stores (other half of the double buffer) follow the MFMAs. Out-of-image pixels: clamped address + select.
  const bool patch_thread = threadIdx.x < 2 * WG_TILES;
  const int it_tile = (threadIdx.x >> 1) & (WG_TILES - 1), it_quad = threadIdx.x & 1;
  const int g_tile = min(tile0 + it_tile, a.ntiles - 1);
  const bool tile_ok = tile0 + it_tile < a.ntiles;
  const int n_img = g_tile / (a.th * a.tw);
  const int rem = g_tile - n_img * (a.th * a.tw);
  const int ty = rem / a.tw, tx = rem - ty * a.tw;
  const int iy0 = 2 * ty - 1, ix0 = 2 * tx - 1;
  const float* xin = a.x + (int64_t)n_img * a.H * a.W * a.cin + it_quad * 4;
  const int64_t per = (int64_t)a.cin * a.cout;
  int64_t poff[16];
  unsigned okmask = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int iy = iy0 + i, ix = ix0 + j;
      const bool ok = tile_ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      okmask |= (ok ? 1u : 0u) << (i * 4 + j);
      poff[i * 4 + j] = ((int64_t)min(max(iy, 0), a.H - 1) * a.W + min(max(ix, 0), a.W - 1)) * a.cin;
    }
  int64_t uoff[16];
  const int ut = threadIdx.x & 127;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int v = q * 128 + ut;                                       // float4 index in the chunk's U block
    const int row = v >> 4, col4 = v & 15;                            // row = xi * CC + c
    const int xi = row / CC, c = row - xi * CC;
    uoff[q] = xi * per + (int64_t)c * a.cout + k0 + col4 * 4;
  }

  f4 d[16];                                                           // staged loads: patch (waves 0,1) or U block (waves 2,3)
  auto load_stage = [&](int c0) {
    if (patch_thread) {
#pragma unroll
      for (int p = 0; p < 16; ++p) d[p] = *reinterpret_cast<const f4*>(xin + poff[p] + c0);
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) d[q] = *reinterpret_cast<const f4*>(a.U + uoff[q] + (int64_t)c0 * a.cout);
    }
  };
  auto store_stage = [&](float* buf) {
    if (patch_thread) {
      // B^T d B, B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]; written as V[xi][c][tile]
      f4 t[16];
#pragma unroll
      for (int p = 0; p < 16; ++p)
        if (!((okmask >> p) & 1u)) d[p] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t[0 * 4 + j] = d[0 * 4 + j] - d[2 * 4 + j];
        t[1 * 4 + j] = d[1 * 4 + j] + d[2 * 4 + j];
        t[2 * 4 + j] = d[2 * 4 + j] - d[1 * 4 + j];
        t[3 * 4 + j] = d[1 * 4 + j] - d[3 * 4 + j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f4 v[4] = {t[i * 4 + 0] - t[i * 4 + 2], t[i * 4 + 1] + t[i * 4 + 2], t[i * 4 + 2] - t[i * 4 + 1],
                         t[i * 4 + 1] - t[i * 4 + 3]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float* dst = buf + ((i * 4 + j) * CC + it_quad * 4) * WG_TILES + it_tile;
          dst[0] = v[j][0];
          dst[WG_TILES] = v[j][1];
          dst[2 * WG_TILES] = v[j][2];
          dst[3 * WG_TILES] = v[j][3];
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) *reinterpret_cast<f4*>(buf + V_FLOATS + (q * 128 + ut) * 4) = d[q];
    }
  };

  f32x16 acc[16];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;

  const int nchunks = a.cin / CC;
  load_stage(0);
  store_stage(lds);
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    const float* buf = lds + (ch & 1) * BUF_FLOATS;
    float* nxt = lds + ((ch + 1) & 1) * BUF_FLOATS;
    const bool more = ch + 1 < nchunks;
    const float* va = buf + kh * WG_TILES + wt * 32 + l31;            // + (xi * CC + 2 s) * 64
    const float* ub = buf + V_FLOATS + kh * WG_K + wk * 32 + l31;     // + (xi * CC + 2 s) * 64
    if (MODE != 2 && more) load_stage((ch + 1) * CC);                 // in flight under this chunk's MFMAs
#pragma unroll
    for (int s = 0; s < CC / 2; ++s) {
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) {
        if (MODE == 1) acc[xi][0] += va[(xi * CC + 2 * s) * WG_TILES] * ub[(xi * CC + 2 * s) * WG_K];
        else acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[(xi * CC + 2 * s) * WG_TILES], ub[(xi * CC + 2 * s) * WG_K],
                                                           acc[xi], 0, 0, 0);
      }
    }
    if (MODE != 2 && more) store_stage(nxt);
    __syncthreads();
  }

  // ---- output transform on the accumulators: Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1]
  const int k = k0 + wk * 32 + l31;
  const float bias = a.bias ? a.bias[k] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;                  // tile of the wave's 32-tile block
    const int gt = tile0 + wt * 32 + row;
    float m[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) m[xi] = acc[xi][r];
    float t0[4], t1[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      t0[s] = m[0 * 4 + s] + m[1 * 4 + s] + m[2 * 4 + s];
      t1[s] = m[1 * 4 + s] - m[2 * 4 + s] - m[3 * 4 + s];
    }
    float y00 = t0[0] + t0[1] + t0[2] + bias, y01 = t0[1] - t0[2] - t0[3] + bias;
    float y10 = t1[0] + t1[1] + t1[2] + bias, y11 = t1[1] - t1[2] - t1[3] + bias;
    if (a.relu) {
      y00 = y00 > 0.f ? y00 : 0.f; y01 = y01 > 0.f ? y01 : 0.f;
      y10 = y10 > 0.f ? y10 : 0.f; y11 = y11 > 0.f ? y11 : 0.f;
    }
    if (gt < a.ntiles) {
      const int n2 = gt / (a.th * a.tw);
      const int rr = gt - n2 * (a.th * a.tw);
      const int oy = 2 * (rr / a.tw), ox = 2 * (rr % a.tw);
      float* yo = a.y + (((int64_t)n2 * a.H + oy) * a.W + ox) * a.cout + k;
      const bool x1 = ox + 1 < a.W, y1 = oy + 1 < a.H;
      yo[0] = y00;
      if (x1) yo[a.cout] = y01;
      if (y1) yo[(int64_t)a.W * a.cout] = y10;
      if (x1 && y1) yo[(int64_t)a.W * a.cout + a.cout] = y11;
    }
  }
}

}  // namespace

CRB_KNOB g_wino_mode = 0;       // measurement builds of the Winograd kernel: 1 = no MFMAs, 2 = no staging
#ifdef CRB_MEASURE
extern "C" int crb_winograd_set_mode(int mode) { g_wino_mode = (mode == 1 || mode == 2) ? mode : 0; return CRB_OK; }
#endif

extern "C" int crb_winograd_supported(int cin, int cout) { return (cin > 0 && cout > 0 && cin % CC == 0 && cout % WG_K == 0) ? 1 : 0; }

extern "C" int64_t crb_winograd_weights_bytes(int cin, int cout) { return (int64_t)16 * cin * cout * 4; }

// g (3,3,Cin,Cout) f32 (ky, kx, input channel, output channel) -> U (16,Cin,Cout)
extern "C" int crb_winograd_weights(const float* g, float* U, int cin, int cout, void* stream) {
  if (!crb_winograd_supported(cin, cout)) return CRB_ERR_UNSUPPORTED;
  const int64_t per = (int64_t)cin * cout;
  hipLaunchKernelGGL(winograd_weights_kernel, dim3(crb_cdiv(per, 256)), dim3(256), 0, (hipStream_t)stream, g, U, cin, cout);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_conv3x3_winograd_nhwc(const float* x, const float* U, float* y, int N, int H, int W, int cin, int cout,
                                         const float* bias, int relu, void* stream) {
  if (!crb_winograd_supported(cin, cout)) return CRB_ERR_UNSUPPORTED;
  if (N <= 0 || H <= 0 || W <= 0) return CRB_ERR_ARG;
  WinoArgs a;
  a.x = x; a.U = U; a.y = y; a.bias = bias;
  a.N = N; a.H = H; a.W = W; a.cin = cin; a.cout = cout; a.relu = relu;
  a.th = (H + 1) / 2; a.tw = (W + 1) / 2;
  const int64_t nt = (int64_t)N * a.th * a.tw;
  if (nt >= (1LL << 31) || (int64_t)N * H * W * (cin > cout ? cin : cout) >= (1LL << 40)) return CRB_ERR_ARG;
  a.ntiles = (int)nt;
  const size_t lds = 2 * BUF_FLOATS * sizeof(float);                  // 128 KB
  auto kern = winograd_f2x2_3x3_kernel<0>;
#ifdef CRB_MEASURE
  if (g_wino_mode == 1) kern = winograd_f2x2_3x3_kernel<1>;
  if (g_wino_mode == 2) kern = winograd_f2x2_3x3_kernel<2>;
#endif
  static bool attr_done = false;
  if (!attr_done || g_wino_mode) {
    CRB_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(crb_cdiv(nt, WG_TILES), cout / WG_K), dim3(256), lds, (hipStream_t)stream, a);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

#endif  // CRB_MEASURE
