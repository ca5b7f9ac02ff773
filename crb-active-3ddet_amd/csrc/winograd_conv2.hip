// 3x3 stride-1 pad-1 convolution on channels_last (NHWC) maps as Winograd F(2x2, 3x3) on the f32 MFMA, second design (round 4)
// (row a7 of SURVEY §8: the BEV backbone's 3x3 convolutions, pcdet/models/backbones_2d/base_bev_backbone.py:24-41).
//
// What the first design (winograd_conv.hip) taught: with all 16 xi of a 32x32 block in one wave (256 accumulators) a SIMD holds
// ONE wave, so the input transform, the LDS stores, the prologue and the output transform all stop the matrix pipe, and every
// 4x4 patch was fetched by the tile that owns it (4x the map through the texture addresser): 1.125 ms where the MFMAs need 0.47.
//
// This design:
//   * workgroup = 512 threads = 8 waves = TWO waves per SIMD, 128 accumulators each (16 xi x 2 blocks of v_mfma_f32_16x16x4_f32):
//     one wave's transforms / LDS traffic / epilogue run under the other's MFMAs.
//   * workgroup tile = 64 tiles (16 tile rows x 4 tile columns of ONE spatial block; tile rows run over the whole batch with
//     the rows of an image rounded up to an even count, so a WAVE's two tile rows are always rows ty, ty + 1 of one image)
//     x 64 output channels; wave = 32 tiles x 16 channels x 16 xi.
//   * raw input and weights come in by LDS-DMA (global_load_lds_dwordx4: no staging registers, a map element crosses the texture
//     path once or twice per workgroup instead of four times). Every wave copies the 6 pixel rows x 10 pixels x 8 channels of
//     ITS two tile rows into a private 2 KB region of the raw buffer (written by this wave's DMA, read by this wave's transform:
//     no other wave waits for it, the copy is issued two chunks ahead and is left in flight across the barrier, vmcnt(2)); the
//     chunk's 32 KB U block is shared (one chunk ahead). The transform V = B^T d B reads the raw rows from LDS (one (tile,
//     channel) per thread and chunk: 16 ds_read_b32, 16 packed-f32 operations, 16 ds_write_b32) - conflict-free through an
//     even/odd pixel-column split of the raw rows.
//   * U is the MFMA's A operand (rows = output channels), V the B operand (columns = tiles): a lane ends with 4 CONSECUTIVE output
//     channels of one tile, so Y = A^T M A happens in registers and goes out as 16-byte stores.
//   * operands are read with ds_read_b128 (two xi x two k-steps per read) from [xi pair][row][channel pair slot][xi parity][2]
//     images whose 16-byte slot index is XORed with 3*((row>>3)&1): conflict-free for the four 16-lane groups a b128 read is
//     served in, without padding (ds_read_b64 pairs get merged into ds_read2st64_b64 by the compiler: half rate and a different
//     banking). U is stored in global memory as that LDS image, chunk by chunk (crb_winograd2_weights): its LDS-DMA is a linear copy.
//   * pipeline per chunk of 8 input channels: 8 stages of [piece of the transform raw(n+1) -> V(n+1) | operand reads | 8 MFMAs on
//     V(n), U(n)], raw(n+3) issued in stage 2, before the last stage's MFMAs [wait for everything but raw(n+3), one barrier, issue
//     U(n+2)]; V, U and raw are double-buffered: 2 x 32 + 2 x 32 + 2 x 16 KB = all 160 KB of LDS.
//   * optional: the slab sums of y and y^2 for the BatchNorm that follows (epilogue, a.stats), the previous layer's BatchNorm +
//     ReLU applied inside the input transform (AFFINE instances, opt-in).
//
// Round 5 - what was built on top of this kernel, measured on the same box against it, and NOT kept (commit
// 4790d45 has the code, profiles/r05_time_winograd2_v1..v4*.txt the numbers):
//   * split tail: the nunits mod CUs units of the last, mostly empty round split along the input channels into S parts, one per
//     workgroup, partial outputs in a workspace, added in part order by the last workgroup to arrive (deterministic, bit-equal
//     reruns, reservation-independent): -2 .. -3 % where a tail exists - and every form of the chunk loop that carried it (item
//     walkers over (unit, chunk range) lists: +2 .. 5 %; a second run of the unchanged pipeline for the part: +2.7 % on the
//     launches with nothing to split, and its two-instance form still +2.7 % for reasons of code placement) cost as much;
//   * first chunk of a unit with C = 0 MFMAs instead of the zeroing pass: +5 % (two alternating chunk bodies);
//   * the output transform overlapped with the next unit's first chunk: does not fit (128 accumulators + 32 transformed values +
//     the chunk's own ~95 registers > 256 per wave at two waves per SIMD: 550 - 680 spill instructions);
//   * a 64 tiles x 128 channels workgroup tile (DESIGN r04 section 8.1): 64 x 128 x 16 accumulators = 512 KB = the CU's whole
//     register file.
// The kernel below is the round-4 kernel; round 5 changed its host side only (per-device launch state, atomic launch sequence,
// reservation honoured by full-size launches only and clamped to half of the CUs, BatchNorm-apply instances removed).
#include <atomic>
#include "crb_common.h"
#include "../../include/crb_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int TB_ROWS = 16, TB_COLS = 4;     // tile block: 16 tile rows x 4 tile columns
constexpr int WG_TILES = TB_ROWS * TB_COLS;  // 64
constexpr int WG_K = 64;                     // output channels per workgroup
constexpr int CC = 8;                        // input channels per chunk
constexpr int V_FLOATS = 16 * WG_TILES * CC; // 8192 = 32 KB
constexpr int U_FLOATS = 16 * WG_K * CC;     // 8192 = 32 KB
constexpr int RAW_ROW_FLOATS = 2 * (TB_COLS + 1) * CC;   // [parity][5 pixel pairs][8 channels] = 80
constexpr int RAW_WAVE_FLOATS = 512;        // a wave's PRIVATE raw region: the 6 pixel rows of its two tile rows (480 floats) + 8 junk slots
constexpr int RAW_FLOATS = 8 * RAW_WAVE_FLOATS;   // 4096 = 16 KB: 1024 DMA slots of 16 bytes, two per thread
constexpr int LDS_FLOATS = 2 * V_FLOATS + 2 * U_FLOATS + 2 * RAW_FLOATS;
constexpr int NT = 512;

__device__ float g_wino_zero_page[64];       // source of out-of-map pixels (zero-initialised, never written)
// CUs that a long-running kernel on ANOTHER stream holds right now (crb_cu_reservation: the farthest-point sampling of PV-RCNN keeps
// one CU per frame for ~5 ms on its side stream). A persistent launch = one workgroup per CU: with 16 CUs taken, 16 of its 256
// workgroups wait for a CU and run their whole unit range after the others (0.73 -> 0.94 ms per call, measured in the scoring pass
// at 16 frames per batch). The first workgroup of a launch latches the number (one word per launch in a small ring, compare-and-swap:
// every workgroup of the launch sees the same value), the launch spreads its units over gridDim - busy workgroups and the last
// `busy` workgroups - the ones that were still waiting for a CU - exit at once.
__device__ int g_cu_busy = 0;
__device__ unsigned g_cu_latch[64];
__global__ void cu_busy_set_kernel(int v) { __hip_atomic_store(&g_cu_busy, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// operand image of one chunk (V: row = tile, U: row = output channel): float index of (xi, row, channel c of the chunk)
__host__ __device__ __forceinline__ constexpr int img_index(int xi, int row, int c) {
  return (xi >> 1) * (64 * 16) + row * 16 + (((c >> 1) ^ (((row >> 3) & 1) * 3)) << 2) + ((xi & 1) << 1) + (c & 1);
}

// g (3,3,Cin,Cout) -> U in LDS-image order: [cout block of 64][chunk of 8 ci][img_index(xi, co & 63, ci & 7)]
__global__ __launch_bounds__(256) void winograd2_weights_kernel(const float* __restrict__ g, float* __restrict__ U, int cin,
                                                                int cout) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per = (int64_t)cin * cout;
  if (t >= per) return;
  const int ci = (int)(t / cout), co = (int)(t - (int64_t)ci * cout);
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
  const int nch = cin / CC;
  const int cb = co / WG_K, col = co - cb * WG_K;
  float* dst = U + (int64_t)(cb * nch + ci / CC) * U_FLOATS;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float u0 = tmp[r][0], u1 = 0.5f * (tmp[r][0] + tmp[r][1] + tmp[r][2]),
                u2 = 0.5f * (tmp[r][0] - tmp[r][1] + tmp[r][2]), u3 = tmp[r][2];
    dst[img_index(r * 4 + 0, col, ci & 7)] = u0;
    dst[img_index(r * 4 + 1, col, ci & 7)] = u1;
    dst[img_index(r * 4 + 2, col, ci & 7)] = u2;
    dst[img_index(r * 4 + 3, col, ci & 7)] = u3;
  }
}

// the same image straight from an nn.Conv2d weight (Cout,Cin,3,3) with arbitrary element strides (contiguous or channels_last):
// mode 0 = forward (kernel input channels = Cin), mode 1 = input gradient: the convolution dy -> dx has the flipped, transposed
// weights g'[ky][kx][co][ci] = w[co][ci][2-ky][2-kx] (kernel input channels = Cout, output channels = Cin)
__device__ __forceinline__ void weights_conv_one(const float* __restrict__ w, int64_t so, int64_t si, int64_t sky, int64_t skx,
                                                 float* __restrict__ U, int kin, int kout, int mode, int64_t t) {
  if (t >= (int64_t)kin * kout) return;
  const int ci = (int)(t / kout), co = (int)(t - (int64_t)ci * kout);      // kernel-side input / output channel
  float g[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
      g[a][b] = mode == 0 ? w[co * so + ci * si + a * sky + b * skx] : w[ci * so + co * si + (2 - a) * sky + (2 - b) * skx];
  float tmp[4][3];                 // G g
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    tmp[0][b] = g[0][b];
    tmp[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
    tmp[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
    tmp[3][b] = g[2][b];
  }
  const int nch = kin / CC;
  const int cb = co / WG_K, col = co - cb * WG_K;
  float* dst = U + (int64_t)(cb * nch + ci / CC) * U_FLOATS;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    dst[img_index(r * 4 + 0, col, ci & 7)] = tmp[r][0];
    dst[img_index(r * 4 + 1, col, ci & 7)] = 0.5f * (tmp[r][0] + tmp[r][1] + tmp[r][2]);
    dst[img_index(r * 4 + 2, col, ci & 7)] = 0.5f * (tmp[r][0] - tmp[r][1] + tmp[r][2]);
    dst[img_index(r * 4 + 3, col, ci & 7)] = tmp[r][2];
  }
}

__global__ __launch_bounds__(256) void winograd2_weights_conv_kernel(const float* __restrict__ w, int64_t so, int64_t si,
                                                                     int64_t sky, int64_t skx, float* __restrict__ U,
                                                                     int kin, int kout, int mode) {
  weights_conv_one(w, so, si, sky, skx, U, kin, kout, mode, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

// the images of MANY layers in one launch (a training step of the BEV backbone transforms 22 weight tensors, ~10 us per launch each
// for 0.6 - 2.4 MB of output: launch-bound). Block b belongs to the job whose block range holds it.
constexpr int WJ_MAX = 32;
struct Wino2WJob { const float* w; float* U; int64_t so, si, sky, skx; int kin, kout, mode, first_block; };
struct Wino2WJobs { int n; Wino2WJob job[WJ_MAX]; };

__global__ __launch_bounds__(256) void winograd2_weights_conv_multi_kernel(Wino2WJobs jobs) {
  int j = 0;
  while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.job[j + 1].first_block) ++j;       // (wave-uniform, <= 31 steps)
  const Wino2WJob& q = jobs.job[j];
  weights_conv_one(q.w, q.so, q.si, q.sky, q.skx, q.U, q.kin, q.kout, q.mode, (int64_t)(blockIdx.x - q.first_block) * 256 + threadIdx.x);
}

struct Wino2Args {
  const float* x;      // (N,H,W,Cin)
  const float* U;      // crb_winograd2_weights image
  float* y;            // (N,H,W,Cout)
  const float* bias;   // (Cout) or null
  float* stats;        // null, or (2 * spatial blocks, 2, Cout): per (spatial block, half of its 64 tiles) the column sums of y and y^2
                       // over the outputs inside the map - the slab sums crb_bn_relu_forward_partials takes (training: the following
                       // BatchNorm's statistics pass over y disappears)
  // BNB instances (input-gradient launches whose output dz is the gradient w.r.t. relu(batchnorm(bn_y))): a.stats receives the slab sums
  // of dz [z > 0] and dz [z > 0] xhat instead of those of y and y^2 - the BatchNorm backward's reduction pass (bn_partial_kernel<true>,
  // same expressions) rides in this epilogue; bn_y (N,H,W,Cout) is the BatchNorm's input, the four vectors its saved statistics / affine
  const float* bn_y; const float* bn_mean; const float* bn_invstd; const float* bn_gamma; const float* bn_beta;
  int bn_relu;
  const float* affine; // AFFINE instances: (Cin, 2) = per input channel (scale, shift): the kernel convolves relu(scale * x + shift)
  int N, H, W, cin, cout, relu;
  int th, tw;          // tile rows per image rounded UP TO EVEN (a wave's two tile rows never straddle two images; the phantom
                       // row of an odd count is computed and not stored), tiles per row = ceil(W/2)
  int RT;              // tile rows over the batch = N * th
  int tw4;             // tile-column blocks = ceil(tw / 4)
  int nblocks;         // spatial blocks = ceil(RT / 16) * tw4
  int ncb;             // cout / 64
  int persistent;      // 1: gridDim.x workgroups share the units as contiguous ranges; 0: one unit per workgroup
  unsigned seq;        // launch sequence number (24 bits, never 0) for the busy-CU latch; 0 = ignore g_cu_busy
};

// one 16-byte LDS-DMA per lane: LDS destination = wave-uniform base + lane * 16
template <int AUX = 0>
__device__ __forceinline__ void glds16(const float* gsrc, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, AUX);
}

// MODE (measurement builds): 1 = no MFMAs, 2 = no transform (V never written), 3 = no DMA after the prologue (all three: wrong
// results); 4 = correct results + per-workgroup stamps {s_memtime at start, after the prologue, after the chunks (incl. the
// output transforms), cycles parked at the chunk barriers, wall_clock64 at start and end, XCC id, units} in g_wino2_dbg; 5 = every raw slot copies the zero page (no map traffic), 6 = U always from the first chunk (both wrong); 7 / 8 / 9 = nt (non-temporal) LDS-DMA for U / raw / both (correct results, A/B)
__device__ unsigned long long* g_wino2_dbg = nullptr;

// 16 bytes through the scalar cache (wave-uniform address): the bias of a unit must not go through the vector memory counter,
// where waiting for it would also wait for the LDS-DMA in flight
__device__ __forceinline__ f32x4 sload4(const float* p) {
  f32x4 r;
  asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
  return r;
}

// A unit = (spatial block of 16 x 4 tiles, block of 64 output channels); units are numbered with the channel block fastest.
// A workgroup owns a contiguous range of units and runs ALL their chunks as one software pipeline (the DMA of the next unit's
// first chunks goes out under the last chunks of the current one: no prologue per unit). Walking from unit to unit needs no
// division: (cb, bc, R0, n0, ty0) advance incrementally.
struct UnitPos {
  int cb, bc, R0, n0, ty0;     // channel block, tile-column block, first tile row over the batch = image n0, row ty0 of it
};
__device__ __forceinline__ void unit_next(UnitPos& u, const Wino2Args& a) {
  if (++u.cb < a.ncb) return;
  u.cb = 0;
  if (++u.bc < a.tw4) return;
  u.bc = 0;
  u.R0 += TB_ROWS;
  u.ty0 += TB_ROWS;
  while (u.ty0 >= a.th) { u.ty0 -= a.th; ++u.n0; }
}

// AFFINE: the input of the convolution is relu(scale[c] * x + shift[c]) (the previous layer's BatchNorm + ReLU, never written to
// memory), zero outside the map like any padded input: applied by the input transform to the 16 values it reads, times a 0 / 1
// mask of the patch positions inside the map. The 8 (scale, shift) pairs of a chunk travel in four of the eight junk slots of the
// wave's second raw DMA instruction: no extra instruction, no extra counter to wait for.
template <int MODE, bool AFFINE = false, bool BNB = false>
__global__ __launch_bounds__(NT, 2) void winograd2_kernel(Wino2Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const Vb = lds;
  float* const Ub = lds + 2 * V_FLOATS;
  float* const Rb = lds + 2 * V_FLOATS + 2 * U_FLOATS;
  constexpr int U_AUX = (MODE == 7 || MODE == 9) ? 2 : 0, RAW_AUX = (MODE == 8 || MODE == 9) ? 2 : 0;   // A/B: nt (streaming) DMA
  const int T = threadIdx.x, lane = T & 63, wave = __builtin_amdgcn_readfirstlane(T >> 6);     // (wave: a scalar for the compiler)
  unsigned long long stamp[6], tm_stage06 = 0, tm_stage7 = 0, tm_book = 0, tm_epi = 0;
  if (MODE == 4) { stamp[0] = __builtin_amdgcn_s_memtime(); stamp[4] = wall_clock64(); stamp[3] = 0; }

  // ---- the unit range of this workgroup
  const int nunits = a.nblocks * a.ncb;
  int u_first, u_end;
  if (a.persistent) {
    int G = gridDim.x;
    if (a.seq) {
      if (T == 0) {
        unsigned* L = g_cu_latch + (a.seq & 63u);
        unsigned v = __hip_atomic_load(L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), mine;
        for (;;) {
          if ((v >> 8) == a.seq) { mine = v & 255u; break; }
          const int b = __hip_atomic_load(&g_cu_busy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned busy = (unsigned)min(max(b, 0), 255);
          const unsigned seen = atomicCAS(L, v, (a.seq << 8) | busy);
          if (seen == v) { mine = busy; break; }
          v = seen;
        }
        Rb[6 * RAW_ROW_FLOATS] = __uint_as_float(mine);         // (junk tail of wave 0's raw region: no DMA has been issued yet)
      }
      __syncthreads();
      const int busy = (int)__float_as_uint(Rb[6 * RAW_ROW_FLOATS]);
      __syncthreads();
      G = max(1, (int)gridDim.x - __builtin_amdgcn_readfirstlane(busy));
      if ((int)blockIdx.x >= G) return;
    }
    u_first = (int)((int64_t)blockIdx.x * nunits / G);
    u_end = (int)((int64_t)(blockIdx.x + 1) * nunits / G);
  } else {
    // consecutive workgroup ids alternate XCDs (id % 8): the channel blocks of one spatial block run back to back on ONE XCD, so
    // that the later ones read the input block from that L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tbl = slot / a.ncb, cb = slot - tbl * a.ncb;
    const int tb = tbl * 8 + xcd;
    u_first = tb < a.nblocks ? tb * a.ncb + cb : 0;
    u_end = tb < a.nblocks ? u_first + 1 : 0;
  }
  if (u_first >= u_end) return;
  const int nch = a.cin / CC;
  const int total = (u_end - u_first) * nch;      // chunks of this workgroup
  UnitPos first;
  {
    const int tb = u_first / a.ncb;
    first.cb = u_first - tb * a.ncb;
    const int br = tb / a.tw4;
    first.bc = tb - br * a.tw4;
    first.R0 = br * TB_ROWS;
    first.n0 = first.R0 / a.th;
    first.ty0 = first.R0 - first.n0 * a.th;
  }
  // image and tile row inside the image of tile row t of the block at u (rows past the batch: the last valid row)
  auto row_of = [&](const UnitPos& u, int t, int& n, int& ty) {
    t = min(t, a.RT - 1 - u.R0);
    n = u.n0;
    ty = u.ty0 + t;
    while (ty >= a.th) { ty -= a.th; ++n; }
    return t;
  };

  // ---- transform role: one (tile, channel) per thread and chunk
  const int t_ch = T & 7, t_tile = T >> 3, t_tr = t_tile >> 2, t_tc = t_tile & 3;
  const int v_off = img_index(0, t_tile, t_ch);
  // the wave's tiles are tile rows 2 wave, 2 wave + 1 of the block; its raw rows live in its OWN region of the raw buffer (local row
  // r = pixel row 2 ty - 1 + r of the first tile row's image; the second tile row starts at local row 2): written by this wave's
  // DMA, read by this wave's transform - no other wave waits for them
  const int raw_off = wave * RAW_WAVE_FLOATS + (2 * (t_tr & 1)) * RAW_ROW_FLOATS + t_tc * CC + t_ch;
  const int sb_off = wave * RAW_WAVE_FLOATS + 6 * RAW_ROW_FLOATS + 2 * t_ch;     // AFFINE: (scale, shift) of the thread's channel
  // AFFINE: mk[i][h] = 1 where patch row i and columns 2 h, 2 h + 1 are inside the map (per unit of the transform, which runs one
  // chunk ahead of the MFMAs)
  UnitPos tu = first;
  int tc = 0;
  f32x2 mk[4][2];
  auto t_setup = [&]() {
    int ty = tu.ty0 + t_tr;
    while (ty >= a.th) ty -= a.th;
    const int x0 = 2 * (tu.bc * TB_COLS + t_tc) - 1;
    float cv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cv[j] = (x0 + j >= 0 && x0 + j < a.W) ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int yy = 2 * ty - 1 + i;
      const float rv = (yy >= 0 && yy < a.H) ? 1.f : 0.f;
      mk[i][0] = (f32x2){rv * cv[0], rv * cv[1]};
      mk[i][1] = (f32x2){rv * cv[2], rv * cv[3]};
    }
  };
  if (AFFINE) t_setup();
  auto t_advance = [&]() {
    if (!AFFINE) return;
    if (++tc < nch) return;
    tc = 0;
    const int R0 = tu.R0, bc = tu.bc;
    unit_next(tu, a);
    if (tu.R0 != R0 || tu.bc != bc) t_setup();
  };

  // ---- DMA role. raw: slots q = T, T + 512 (16 bytes each): q -> (raw row, parity, pixel pair, channel half); runs three chunks
  //      ahead of the MFMAs. U: 4 x 16 bytes per thread, two chunks ahead.
  UnitPos ru = first, uu = first;
  int rc = 0, uc = 0, r_issued = 0, u_issued = 0;
  const float* rsrc[2];
  int rstep[2];
  // sources of the two slots of this thread for the block at u. touch = false: the 16 bytes a slot copies (chunk 0), step = one
  // chunk; touch = true (line prefetch): ONE lane per pixel (the channel-half-0 slot) gets the pixel's first channel, every other
  // lane the zero page
  auto slot_sources = [&](const UnitPos& u, const float* (&src)[2], int (&step)[2], bool touch) {
    // the wave's first tile row: image n, row ty of it (th is even: both tile rows of the wave are rows ty, ty + 1 of image n)
    int n = u.n0, ty = u.ty0 + 2 * wave;
    while (ty >= a.th) { ty -= a.th; ++n; }
    const bool rows_in_batch = u.R0 + 2 * wave < a.RT;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int q = lane + 64 * s;                           // slot of the wave's region: (local row 0..5 | junk, parity, pair, half)
      const int rr = q / 20, rem = q - rr * 20;
      const int par = rem / 10, rem2 = rem - par * 10;
      const int xh = rem2 >> 1, half = rem2 & 1;
      const int yy = 2 * ty - 1 + rr;
      const int xx = 8 * u.bc - 1 + 2 * xh + par;
      const bool ok = rows_in_batch && rr < 6 && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W && MODE != 5 && !(touch && half);
      src[s] = ok ? a.x + (((int64_t)n * a.H + yy) * a.W + xx) * a.cin + half * 4 : g_wino_zero_page;
      step[s] = ok ? CC : 0;
      if (AFFINE && q >= 120 && q < 124) {                   // junk slots 0..3: channels 2 p, 2 p + 1 of the chunk as (s, b, s, b)
        src[s] = a.affine + (q - 120) * 4;
        step[s] = 2 * CC;
      }
    }
  };
  auto r_setup = [&]() { slot_sources(ru, rsrc, rstep, false); };
  r_setup();
  const float* usrc = a.U + (int64_t)uu.cb * nch * U_FLOATS + T * 4;
  auto issue_raw = [&]() {                                    // chunk r_issued -> raw buffer r_issued & 1
    float* buf = Rb + (r_issued & 1) * RAW_FLOATS + wave * RAW_WAVE_FLOATS;
    glds16<RAW_AUX>(rsrc[0], buf);
    glds16<RAW_AUX>(rsrc[1], buf + 256);                     // lanes 56..63: junk slots (zero page -> the tail of the region)
    ++r_issued;
  };
  auto r_advance = [&]() {                                    // after issue_raw: move the sources to the next chunk
    if (++rc < nch) {
      rsrc[0] += rstep[0];
      rsrc[1] += rstep[1];
      return;
    }
    rc = 0;
    const int R0 = ru.R0, bc = ru.bc;
    unit_next(ru, a);
    if (ru.R0 != R0 || ru.bc != bc) r_setup();
    else {                                                    // same spatial block, next channel block: back to channel 0
      rsrc[0] -= (nch - 1) * rstep[0];
      rsrc[1] -= (nch - 1) * rstep[1];
    }
  };
  auto issue_u = [&]() {                                      // chunk u_issued -> U buffer u_issued & 1
    float* buf = Ub + (u_issued & 1) * U_FLOATS;
#pragma unroll
    for (int k = 0; k < 4; ++k) glds16<U_AUX>(usrc + k * NT * 4, buf + (k * NT + wave * 64) * 4);
    ++u_issued;
  };
  auto u_advance = [&]() {
    if (MODE == 6) return;
    if (++uc < nch) { usrc += U_FLOATS; return; }
    uc = 0;
    unit_next(uu, a);
    usrc = a.U + (int64_t)uu.cb * nch * U_FLOATS + T * 4;
  };

  // ---- MFMA role: wave = 32 tiles (wt) x 16 output channels (wk) x 16 xi
  const int wt = wave >> 2, wk = wave & 3;
  const int l15 = lane & 15, kq = lane >> 4;
  const int a_off = img_index(0, wk * 16 + l15, 2 * kq);        // U image: A operand, rows = output channels
  const int b_off = img_index(0, wt * 32 + l15, 2 * kq);        // V image: B operand, columns = tiles (second block + 16 rows)
  f32x4 acc[16][2];
  auto acc_clear = [&]() {
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
      acc[xi][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc[xi][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  acc_clear();
  auto wait_all_and_barrier = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };


  // ---- output transform of a finished unit on the accumulators: Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1]; lane = tile l15 of
  //      the wave's block, 4 consecutive output channels; then the accumulators start the next unit at zero
  UnitPos eu = first;
  int ec = 0;
  auto unit_epilogue = [&]() __attribute__((always_inline)) {
    const int k = eu.cb * WG_K + wk * 16 + 4 * kq;
    f32x4 bias = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!BNB && a.bias) {
      const float* bp = a.bias + eu.cb * WG_K + __builtin_amdgcn_readfirstlane(wk) * 16;
      const f32x4 b0 = sload4(bp), b1 = sload4(bp + 4), b2 = sload4(bp + 8), b3 = sload4(bp + 12);
#pragma unroll
      for (int e = 0; e < 4; ++e) bias[e] = kq == 0 ? b0[e] : kq == 1 ? b1[e] : kq == 2 ? b2[e] : b3[e];
    }
    f32x4 s1 = (f32x4){0.f, 0.f, 0.f, 0.f}, s2 = (f32x4){0.f, 0.f, 0.f, 0.f};     // a.stats: this lane's sums over its 8 outputs
    f32x4 bmu, bis, bga, bbe, bv[2][4];
    if constexpr (BNB) {
      // the BatchNorm's input at this lane's 8 output positions: requested before the output transform, used after its stores
      bmu = *reinterpret_cast<const f32x4*>(a.bn_mean + k); bis = *reinterpret_cast<const f32x4*>(a.bn_invstd + k);
      bga = *reinterpret_cast<const f32x4*>(a.bn_gamma + k); bbe = *reinterpret_cast<const f32x4*>(a.bn_beta + k);
#pragma unroll
      for (int tbk = 0; tbk < 2; ++tbk) {
        const int tile = wt * 32 + tbk * 16 + l15;
        const int Rg = eu.R0 + (tile >> 2);
        const int tx = eu.bc * TB_COLS + (tile & 3);
        int n2, ty2;
        row_of(eu, tile >> 2, n2, ty2);
        const int oy = 2 * ty2, ox = 2 * tx;
        const bool in = Rg < a.RT && tx < a.tw && oy < a.H;
        const bool x1 = ox + 1 < a.W, y1 = oy + 1 < a.H;
        const float* yp = a.bn_y + (((int64_t)n2 * a.H + oy) * a.W + ox) * a.cout + k;
        const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        bv[tbk][0] = in ? *reinterpret_cast<const f32x4*>(yp) : zero4;
        bv[tbk][1] = (in && x1) ? *reinterpret_cast<const f32x4*>(yp + a.cout) : zero4;
        bv[tbk][2] = (in && y1) ? *reinterpret_cast<const f32x4*>(yp + (int64_t)a.W * a.cout) : zero4;
        bv[tbk][3] = (in && x1 && y1) ? *reinterpret_cast<const f32x4*>(yp + (int64_t)a.W * a.cout + a.cout) : zero4;
      }
    }
#pragma unroll
    for (int tbk = 0; tbk < 2; ++tbk) {
      const int tile = wt * 32 + tbk * 16 + l15;
      const int Rg = eu.R0 + (tile >> 2);
      const int tx = eu.bc * TB_COLS + (tile & 3);
      int n2, ty2;
      row_of(eu, tile >> 2, n2, ty2);
      f32x4 t0[4], t1[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        t0[s] = acc[0 * 4 + s][tbk] + acc[1 * 4 + s][tbk] + acc[2 * 4 + s][tbk];
        t1[s] = acc[1 * 4 + s][tbk] - acc[2 * 4 + s][tbk] - acc[3 * 4 + s][tbk];
      }
      f32x4 y00 = t0[0] + t0[1] + t0[2] + bias, y01 = t0[1] - t0[2] - t0[3] + bias;
      f32x4 y10 = t1[0] + t1[1] + t1[2] + bias, y11 = t1[1] - t1[2] - t1[3] + bias;
      if (a.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y00[e] = fmaxf(y00[e], 0.f); y01[e] = fmaxf(y01[e], 0.f);
          y10[e] = fmaxf(y10[e], 0.f); y11[e] = fmaxf(y11[e], 0.f);
        }
      }
      if (Rg < a.RT && tx < a.tw && 2 * ty2 < a.H) {                // (2 ty2 >= H: the phantom tile row of an odd row count)
        const int oy = 2 * ty2, ox = 2 * tx;
        float* yo = a.y + (((int64_t)n2 * a.H + oy) * a.W + ox) * a.cout + k;
        const bool x1 = ox + 1 < a.W, y1 = oy + 1 < a.H;
        *reinterpret_cast<f32x4*>(yo) = y00;
        if (x1) *reinterpret_cast<f32x4*>(yo + a.cout) = y01;
        if (y1) *reinterpret_cast<f32x4*>(yo + (int64_t)a.W * a.cout) = y10;
        if (x1 && y1) *reinterpret_cast<f32x4*>(yo + (int64_t)a.W * a.cout + a.cout) = y11;
        if constexpr (BNB) {
          // BatchNorm backward sums of the positions just written, bn_partial_kernel<true>'s expressions: xhat = (y - mean) invstd,
          // d = dz [gamma xhat + beta > 0], sums of d and d xhat; positions outside the map contribute nothing
          const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
          const f32x4 v00 = bv[tbk][0], v01 = bv[tbk][1], v10 = bv[tbk][2], v11 = bv[tbk][3];
          auto term = [&](const f32x4 v, f32x4 d, bool in) __attribute__((always_inline)) {
            const f32x4 xh = (v - bmu) * bis;
            if (a.bn_relu) {
              const f32x4 z = bga * xh + bbe;
#pragma unroll
              for (int e = 0; e < 4; ++e) d[e] = z[e] > 0.f ? d[e] : 0.f;
            }
            if (!in) d = zero4;
            s1 = s1 + d;
            s2 = s2 + d * xh;
          };
          term(v00, y00, true); term(v01, y01, x1); term(v10, y10, y1); term(v11, y11, x1 && y1);
        } else if (a.stats) {                                         // fixed order: (0,0), (0,1), (1,0), (1,1) of tile block 0, then 1
          const float m01 = x1 ? 1.f : 0.f, m10 = y1 ? 1.f : 0.f, m11 = (x1 && y1) ? 1.f : 0.f;
          s1 = s1 + y00; s2 = s2 + y00 * y00;
          s1 = s1 + y01 * m01; s2 = s2 + (y01 * y01) * m01;
          s1 = s1 + y10 * m10; s2 = s2 + (y10 * y10) * m10;
          s1 = s1 + y11 * m11; s2 = s2 + (y11 * y11) * m11;
        }
      }
    }
    if (a.stats) {
      // sum over the 16 tiles of the lane row (rotations inside the row: every lane ends with the total, fixed order), one lane per
      // row writes its four channels: slab = (spatial block, wt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float u = s1[e], v = s2[e];
#define CRB_ROW_ROR_ADD(x, ctrl) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, false))
        CRB_ROW_ROR_ADD(u, 0x128); CRB_ROW_ROR_ADD(v, 0x128);       // row_ror:8
        CRB_ROW_ROR_ADD(u, 0x124); CRB_ROW_ROR_ADD(v, 0x124);
        CRB_ROW_ROR_ADD(u, 0x122); CRB_ROW_ROR_ADD(v, 0x122);
        CRB_ROW_ROR_ADD(u, 0x121); CRB_ROW_ROR_ADD(v, 0x121);
#undef CRB_ROW_ROR_ADD
        s1[e] = u;
        s2[e] = v;
      }
      if (l15 == 0) {
        const int64_t blk = (int64_t)(eu.R0 / TB_ROWS) * a.tw4 + eu.bc;
        float* so = a.stats + ((blk * 2 + wt) * 2) * a.cout + k;
        *reinterpret_cast<f32x4*>(so) = s1;
        *reinterpret_cast<f32x4*>(so + a.cout) = s2;
      }
    }
    acc_clear();
    unit_next(eu, a);
  };

  // ---- one chunk as 8 stages (one xi pair each), pinned by sched_barriers so that a wave's instruction stream alternates
  //      [a piece of the transform of the next chunk | 3 operand reads of the NEXT pair | 8 MFMAs]: the LDS / VALU work of the
  //      transform sits in the shadow of the wave's own MFMAs (and of the SIMD's other wave), not in front of the whole chunk.
  //      transform pieces: stage 0 raw reads d (16), stages 1-2 column pass t = B^T d, stages 3-6 one row of t B + its 4 stores.
  // packed f32: a register pair holds two columns (j, j + 1) of a patch row; the column pass is 8 v_pk_add_f32, a row of
  // t B is two more with operand selects / negations folded in (32 scalar adds before: every vector instruction issued beside
  // the MFMAs costs matrix-pipe time). Bit-identical to the scalar form (a - b = a + (-b)).
  f32x2 dp[4][2], tp[4][2];
  auto t_load = [&](const float* raw) {
    const float* p = raw + raw_off;
    if (MODE == 12) {                   // measurement: no raw reads (the transform works on whatever the registers hold)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) asm volatile("" : "+v"(dp[i][h]));
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h)       // columns 2 h, 2 h + 1: pixel pair h of parity 0 / 1
        dp[i][h] = (f32x2){p[i * RAW_ROW_FLOATS + h * CC], p[i * RAW_ROW_FLOATS + (RAW_ROW_FLOATS / 2) + h * CC]};
    if (AFFINE) {                     // (in this stage: the masks belong to the transform's unit, which stage 1 moves on)
      const f32x2 sbv = *reinterpret_cast<const f32x2*>(raw + sb_off);
      const f32x2 ss = (f32x2){sbv[0], sbv[0]}, bb = (f32x2){sbv[1], sbv[1]}, zero = (f32x2){0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) dp[i][h] = __builtin_elementwise_max(__builtin_elementwise_fma(dp[i][h], ss, bb), zero) * mk[i][h];
    }
  };
  auto t_cols = [&](int h) {        // B^T d, B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1], columns 2 h and 2 h + 1
    tp[0][h] = dp[0][h] - dp[2][h];
    tp[1][h] = dp[1][h] + dp[2][h];
    tp[2][h] = dp[2][h] - dp[1][h];
    tp[3][h] = dp[1][h] - dp[3][h];
  };
  auto t_row = [&](float* V, int i) {
    float* o = V + v_off;
    f32x2 o01, o23;                 // (t0 - t2, t1 + t2), (t2 - t1, t1 - t3) with (t0, t1) = tp[i][0], (t2, t3) = tp[i][1]
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(o01) : "v"(tp[i][0]), "v"(tp[i][1]));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(o23) : "v"(tp[i][0]), "v"(tp[i][1]));
    if (MODE == 13) {                   // measurement: no V stores (results computed, kept alive, not written)
      asm volatile("" :: "v"(o01), "v"(o23));
      return;
    }
    o[img_index(i * 4 + 0, 0, 0)] = o01[0];
    o[img_index(i * 4 + 1, 0, 0)] = o01[1];
    o[img_index(i * 4 + 2, 0, 0)] = o23[0];
    o[img_index(i * 4 + 3, 0, 0)] = o23[1];
  };
  auto transform = [&](const float* raw, float* V) {
    t_load(raw);
    t_cols(0);
    t_cols(1);
#pragma unroll
    for (int i = 0; i < 4; ++i) t_row(V, i);
  };
  f32x4 ua[2], v0[2], v1[2];
  auto op_read = [&](const float* V, const float* U, int xp, int slot) {
    ua[slot] = *reinterpret_cast<const f32x4*>(U + xp * 1024 + a_off);
    v0[slot] = *reinterpret_cast<const f32x4*>(V + xp * 1024 + b_off);
    v1[slot] = *reinterpret_cast<const f32x4*>(V + xp * 1024 + b_off + 16 * 16);
  };
  // the 8 MFMAs of an xi pair in k-step-major order: the two MFMAs of one accumulator are four instructions apart (two apart,
  // a wave that has the matrix pipe to itself waits for the first one's result: measured 768 -> 748 us per 128->128 call)
  auto mfma_pair = [&](int xp, int slot) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int xi = 2 * xp + h;
        if (MODE == 1) {
          acc[xi][0][0] += ua[slot][2 * h + e] * v0[slot][2 * h + e];
          acc[xi][1][0] += ua[slot][2 * h + e] * v1[slot][2 * h + e];
        } else {
          acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[slot][2 * h + e], v0[slot][2 * h + e], acc[xi][0], 0, 0, 0);
          acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[slot][2 * h + e], v1[slot][2 * h + e], acc[xi][1], 0, 0, 0);
        }
      }
  };
  // chunk g of the workgroup. DO_U / DO_RAW / DO_T: compile-time (integral_constant) switches of the peeled tail
  auto chunk = [&](int g, auto do_u, auto do_raw, auto do_t) {
    const int cur = g & 1, nxt = cur ^ 1;
    const float* V = Vb + cur * V_FLOATS;
    const float* U = Ub + cur * U_FLOATS;
    float* Vn = Vb + nxt * V_FLOATS;
    unsigned long long tm0 = 0, tm1 = 0;
    if (MODE == 4) tm0 = __builtin_amdgcn_s_memtime();
    constexpr bool T_ON = decltype(do_t)::value && MODE != 2;       // (the first pair's operands were read by the previous chunk)
#pragma unroll
    for (int xp = 0; xp < 8; ++xp) {
      if (T_ON) {                                                 // LDS stores ahead of the next pair's reads: the wait for the
        if (xp == 0) t_load(Rb + nxt * RAW_FLOATS);               // reads (in-order LDS) then never waits for a younger store
        if (xp == 1) t_cols(0);
        if (xp == 2) t_cols(1);
        if (xp >= 3 && xp < 7) t_row(Vn, xp - 3);
      }
      if (xp == 1) {
        // bookkeeping here, under this stage's MFMAs (at the end of a chunk it cost ~280 cycles with the matrix pipe idle): the
        // DMA sources move on from what the PREVIOUS chunk's last stage issued, the transform from what stage 0 just read
        if (MODE != 3 && MODE != 11) { u_advance(); r_advance(); }      // (MODE 11: sources never move on: wrong results, timing of the bookkeeping)
        if (MODE != 11) t_advance();
      }
      // raw block of chunk g + 3 into the wave's own region of the buffer stage 0 just read (lgkmcnt(0): those reads have
      // returned). Private regions: no other wave's progress matters, the copy has two chunks to land, and the wait in front of
      // the barrier below leaves it in flight (vmcnt(2): loads return in order, these two instructions are the youngest)
      if (xp == 2 && MODE != 3 && decltype(do_raw)::value) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_raw();
      }
      if (xp < 7) op_read(V, U, xp + 1, (xp + 1) & 1);
      // the barrier sits BEFORE the last pair's MFMAs (their operands are in registers, V(g+1) is complete): the waves meet
      // with 8 MFMAs each still to issue, so the matrix pipe keeps running while the DMA of chunks g+2 (U) / g+3 (raw) - into
      // the buffers nobody reads any more - and the next chunk's first reads go out
      if (xp == 7) {
        unsigned long long t0 = 0;
        if (MODE == 4) t0 = __builtin_amdgcn_s_memtime();         // time parked at the wait + barrier (stamp[3] accumulates)
        if (decltype(do_raw)::value && MODE != 3) {
          asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        } else {
          wait_all_and_barrier();
        }
        if (MODE == 4) {
          tm1 = __builtin_amdgcn_s_memtime();
          stamp[3] += tm1 - t0;
          tm_stage06 += t0 - tm0;
        }
        if (MODE != 3 && MODE != 10 && decltype(do_u)::value) issue_u();      // (MODE 10: no U copies in the loop, wrong results)
      }
      mfma_pair(xp, xp & 1);
      // the next chunk's first operands go out right behind the last MFMAs (V(g+1), U(g+1) are valid after the barrier): their
      // latency and the bookkeeping below run under those MFMAs instead of in front of the next chunk's. (Measured the other way
      // round - reads, MFMAs, then the DMA of U(g+2) - +3.5 %: the copy loses lead time it needs before the next barrier.)
      if (xp == 7 && decltype(do_t)::value) op_read(Vn, Ub + nxt * U_FLOATS, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 4) { tm0 = __builtin_amdgcn_s_memtime(); tm_stage7 += tm0 - tm1; }
    if (MODE == 4) { tm1 = __builtin_amdgcn_s_memtime(); tm_book += tm1 - tm0; }
    if (++ec == nch) { ec = 0; unit_epilogue(); }
    if (MODE == 4) tm_epi += __builtin_amdgcn_s_memtime() - tm1;
  };
  using std::true_type;
  using std::false_type;

  // ---- prologue (once per workgroup): raw(0), raw(1), U(0) in one round trip, raw(0) -> V(0), then U(1), raw(2) go out: chunk g
  //      sends raw(g+3) in its stage 2 (two chunks of lead) and U(g+2) after its barrier (one chunk: U comes from the XCD's L2)
  issue_raw(); r_advance();
  if (total > 1) { issue_raw(); r_advance(); }
  issue_u(); u_advance();
  wait_all_and_barrier();
  transform(Rb, Vb); t_advance();
  wait_all_and_barrier();
  if (total > 1) issue_u();                   // (chunk 0 moves the sources on in its stage 1)
  if (total > 2) issue_raw();
  op_read(Vb, Ub, 0, 0);
  if (MODE == 4) stamp[1] = __builtin_amdgcn_s_memtime();

  // ---- chunks: the steady state is one body without DMA / transform conditions, the last three chunks are peeled
  int g = 0;
  for (; g + 3 < total; ++g) chunk(g, true_type{}, true_type{}, true_type{});
  if (g + 2 < total) { chunk(g, true_type{}, false_type{}, true_type{}); ++g; }        // 1 .. 3 chunks left
  if (g + 1 < total) { chunk(g, false_type{}, false_type{}, true_type{}); ++g; }
  chunk(g, false_type{}, false_type{}, false_type{});
  if (MODE == 4) stamp[2] = __builtin_amdgcn_s_memtime();
  if (MODE == 4 && g_wino2_dbg && T == 0) {
    stamp[5] = wall_clock64();
    unsigned long long* o = g_wino2_dbg + (int64_t)blockIdx.x * 16;
#pragma unroll
    for (int i = 0; i < 6; ++i) o[i] = stamp[i];
    o[6] = __builtin_amdgcn_s_getreg(0x14 | (0 << 6) | (3 << 11));  // XCC_ID (hwreg 20, 4 bits)
    o[7] = (unsigned long long)(u_end - u_first);
    o[8] = tm_stage06; o[9] = tm_stage7; o[10] = tm_book; o[11] = tm_epi;
  }
}

}  // namespace

CRB_KNOB g_wino2_persistent = 1; // 1: one workgroup per CU over a range of units (measured 4 - 8 % faster); 0: one unit per workgroup
CRB_KNOB g_wino2_mode [[maybe_unused]] = 0;      // measurement builds: 1 = no MFMAs, 2 = no transform, 3 = no DMA in the loop
#ifdef CRB_MEASURE
extern "C" int crb_winograd2_set_mode(int mode) { g_wino2_mode = (mode >= 1 && mode <= 13) ? mode : 0; return CRB_OK; }
extern "C" int crb_winograd2_set_persistent(int on) { g_wino2_persistent = on ? 1 : 0; return CRB_OK; }
// mode 4: 16 uint64 per workgroup (device buffer of the caller, NULL = off)
extern "C" int crb_winograd2_set_debug(void* dev_buf) {
  unsigned long long* p = (unsigned long long*)dev_buf;
  CRB_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wino2_dbg), &p, sizeof(p)));
  return CRB_OK;
}
#endif

// H >= 5: a block of 16 tile rows crosses at most 5 image boundaries (44 raw rows = 880 slots; slots 896.. are the junk corner)
extern "C" int crb_winograd2_supported(int cin, int cout, int H, int W) {
  return (cin > 0 && cout > 0 && cin % CC == 0 && cout % WG_K == 0 && H >= 5 && W >= 1) ? 1 : 0;
}

extern "C" int64_t crb_winograd2_weights_bytes(int cin, int cout) { return (int64_t)16 * cin * cout * 4; }

// g (3,3,Cin,Cout) f32 (ky, kx, input channel, output channel) -> U image of crb_conv3x3_winograd2_nhwc
extern "C" int crb_winograd2_weights(const float* g, float* U, int cin, int cout, void* stream) {
  if (!crb_winograd2_supported(cin, cout, 5, 1)) return CRB_ERR_UNSUPPORTED;
  const int64_t per = (int64_t)cin * cout;
  hipLaunchKernelGGL(winograd2_weights_kernel, dim3(crb_cdiv(per, 256)), dim3(256), 0, (hipStream_t)stream, g, U, cin, cout);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

// w = nn.Conv2d weight (Cout,Cin,3,3) f32 with element strides (so, si, sky, skx); mode 0: U of the forward convolution
// (Cin -> Cout), mode 1: U of the input-gradient convolution (Cout -> Cin)
extern "C" int crb_winograd2_weights_conv(const float* w, int64_t so, int64_t si, int64_t sky, int64_t skx, float* U, int conv_cin,
                                          int conv_cout, int mode, void* stream) {
  const int kin = mode ? conv_cout : conv_cin, kout = mode ? conv_cin : conv_cout;
  if (!crb_winograd2_supported(kin, kout, 5, 1)) return CRB_ERR_UNSUPPORTED;
  const int64_t per = (int64_t)kin * kout;
  hipLaunchKernelGGL(winograd2_weights_conv_kernel, dim3(crb_cdiv(per, 256)), dim3(256), 0, (hipStream_t)stream, w, so, si, sky, skx,
                     U, kin, kout, mode ? 1 : 0);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

// n weight tensors in one launch: w[j] (Cout_j, Cin_j, 3, 3) with element strides strides[4 j .. 4 j + 3] = (so, si, sky, skx), U[j] its
// image for mode[j] (0 forward, 1 input gradient). n <= 32.
extern "C" int crb_winograd2_weights_conv_multi(int n, const float* const* w, const int64_t* strides, float* const* U,
                                                const int32_t* conv_cin, const int32_t* conv_cout, const int32_t* mode, void* stream) {
  if (n < 0 || n > WJ_MAX || (n > 0 && (!w || !strides || !U || !conv_cin || !conv_cout || !mode))) return CRB_ERR_ARG;
  if (n == 0) return CRB_OK;
  Wino2WJobs jobs;
  jobs.n = n;
  int64_t blocks = 0;
  for (int j = 0; j < n; ++j) {
    const int kin = mode[j] ? conv_cout[j] : conv_cin[j], kout = mode[j] ? conv_cin[j] : conv_cout[j];
    if (!w[j] || !U[j]) return CRB_ERR_ARG;
    if (!crb_winograd2_supported(kin, kout, 5, 1)) return CRB_ERR_UNSUPPORTED;
    jobs.job[j] = Wino2WJob{w[j], U[j], strides[4 * j], strides[4 * j + 1], strides[4 * j + 2], strides[4 * j + 3], kin, kout,
                            mode[j] ? 1 : 0, (int)blocks};
    blocks += crb_cdiv((int64_t)kin * kout, 256);
    if (blocks >= (1LL << 30)) return CRB_ERR_ARG;
  }
  hipLaunchKernelGGL(winograd2_weights_conv_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, jobs);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

// per-device launch state (a process may drive several devices from several threads)
namespace {
constexpr int MAX_DEV = 64;
std::atomic<int> g_dev_cus[MAX_DEV];
std::atomic<unsigned> g_dev_attr[MAX_DEV];                    // bit m: hipFuncSetAttribute done for kernel instance m on this device
std::atomic<unsigned> g_dev_seq[MAX_DEV];                     // launch sequence PER DEVICE: the latch ring (64 slots) lives in that device's memory,
                                                             // two launches share a slot only if 64 launches to the SAME device lie between them

int device_cus(int* dev_out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return -1;
  *dev_out = dev;
  int n = g_dev_cus[dev].load(std::memory_order_relaxed);
  if (n > 0) return n;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
  n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  g_dev_cus[dev].store(n, std::memory_order_relaxed);
  return n;
}
}  // namespace

struct Wino2Bnb { const float* y; const float* mean; const float* invstd; const float* gamma; const float* beta; int relu; };

static int winograd2_launch(const float* x, const float* U, float* y, int N, int H, int W, int cin, int cout, const float* bias,
                            int relu, void* stream, float* stats = nullptr, const Wino2Bnb* bnb = nullptr) {
  if (N <= 0 || H <= 0 || W <= 0) return CRB_ERR_ARG;
  if (!crb_winograd2_supported(cin, cout, H, W)) return CRB_ERR_UNSUPPORTED;
  Wino2Args a;
  a.x = x; a.U = U; a.y = y; a.bias = bias; a.affine = nullptr; a.stats = stats;
  a.bn_y = a.bn_mean = a.bn_invstd = a.bn_gamma = a.bn_beta = nullptr; a.bn_relu = 0;
  if (bnb) {
    a.bn_y = bnb->y; a.bn_mean = bnb->mean; a.bn_invstd = bnb->invstd; a.bn_gamma = bnb->gamma; a.bn_beta = bnb->beta;
    a.bn_relu = bnb->relu;
  }
  a.N = N; a.H = H; a.W = W; a.cin = cin; a.cout = cout; a.relu = relu;
  a.th = (((H + 1) / 2) + 1) & ~1; a.tw = (W + 1) / 2;
  const int64_t rt = (int64_t)N * a.th;
  if (rt >= (1LL << 30) || (int64_t)N * H * W * (cin > cout ? cin : cout) >= (1LL << 40)) return CRB_ERR_ARG;
  a.RT = (int)rt;
  a.tw4 = (a.tw + TB_COLS - 1) / TB_COLS;
  const int64_t nb = (int64_t)((rt + TB_ROWS - 1) / TB_ROWS) * a.tw4;
  if (nb >= (1LL << 26)) return CRB_ERR_ARG;
  a.nblocks = (int)nb;
  a.ncb = cout / WG_K;
  const size_t lds = LDS_FLOATS * sizeof(float);
  int mode = 0;
  auto kern = winograd2_kernel<0>;
#ifdef CRB_MEASURE
  mode = g_wino2_mode;
  if (mode == 1) kern = winograd2_kernel<1>;
  if (mode == 2) kern = winograd2_kernel<2>;
  if (mode == 3) kern = winograd2_kernel<3>;
  if (mode == 4) kern = winograd2_kernel<4>;
  if (mode == 5) kern = winograd2_kernel<5>;
  if (mode == 6) kern = winograd2_kernel<6>;
  if (mode == 7) kern = winograd2_kernel<7>;
  if (mode == 8) kern = winograd2_kernel<8>;
  if (mode == 9) kern = winograd2_kernel<9>;
  if (mode == 10) kern = winograd2_kernel<10>;
  if (mode == 11) kern = winograd2_kernel<11>;
  if (mode == 12) kern = winograd2_kernel<12>;
  if (mode == 13) kern = winograd2_kernel<13>;
#endif
#ifdef CRB_MEASURE
  if (bnb) {
    if (mode != 0) return CRB_ERR_UNSUPPORTED;
    kern = winograd2_kernel<0, false, true>;
    mode = 14;                                                // (attribute bit of this instance)
  }
#else
  if (bnb) return CRB_ERR_UNSUPPORTED;                        // (the BNB instance lives in the measurement library only)
#endif
  int dev = 0;
  const int n_cu = device_cus(&dev);
  if (n_cu <= 0) return CRB_ERR_LAUNCH;
  if (!(g_dev_attr[dev].load(std::memory_order_acquire) & (1u << mode))) {
    CRB_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    g_dev_attr[dev].fetch_or(1u << mode, std::memory_order_release);
  }
  // persistent workgroups: one per CU (all of the LDS each), every one runs a contiguous range of units as one pipeline
  const int64_t units = nb * a.ncb;
  a.persistent = g_wino2_persistent;
  const int64_t grid = a.persistent ? (units < n_cu ? units : n_cu) : ((nb + 7) / 8) * 8 * a.ncb;
  // the busy-CU latch (crb_cu_reservation) is for launches that put a workgroup on EVERY CU: a smaller launch leaves CUs free
  // anyway, and giving up workgroups there would funnel it into a few. The sequence number is per device and
  // atomic (launches from several host threads / to several devices); slot = seq mod 64 of the device's latch ring.
  unsigned seq = (g_dev_seq[dev].fetch_add(1, std::memory_order_relaxed) + 1) & 0xffffffu;
  a.seq = (a.persistent && grid == n_cu) ? (seq ? seq : 1) : 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, (hipStream_t)stream, a);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_conv3x3_winograd2_nhwc(const float* x, const float* U, float* y, int N, int H, int W, int cin, int cout,
                                          const float* bias, int relu, void* stream) {
  return winograd2_launch(x, U, y, N, H, W, cin, cout, bias, relu, stream);
}

// training forward that also writes the slab sums of its output for the BatchNorm that follows: stats (crb_winograd2_stats_slabs, 2,
// Cout) f32 = column sums of y and y^2 per (spatial block of 16 x 4 tiles, half of its tiles), every slab written exactly once
// (no atomics, fixed order inside: bit-reproducible) -> crb_bn_relu_forward_partials(y, n, Cout, stats, slabs, ...)
extern "C" int64_t crb_winograd2_stats_slabs(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  const int64_t th = (((H + 1) / 2) + 1) & ~1, tw4 = ((W + 1) / 2 + TB_COLS - 1) / TB_COLS;
  return 2 * ((N * th + TB_ROWS - 1) / TB_ROWS) * tw4;
}

#ifdef CRB_MEASURE
// VERDICT r04 item 6a, measured (tools/time_wino_bnbwd.py, profiles/r05_time_wino_bnbwd.txt) and NOT adopted: the epilogue costs +52 us
// per 128-channel launch at 16 x 200 x 176 (the reduction pass it replaces: 105 us, the slab reduction 16 us: net 38 us) and +85 us per
// 256-channel launch at 100 x 88 (reduction pass 55 us: a loss) - with the matrix pipe idle, every epilogue instruction is serial time.
// input-gradient launch whose output dz is the gradient w.r.t. relu(batchnorm(bn_y)) of the previous layer: besides dz it writes the
// slab sums (crb_winograd2_stats_slabs, 2, Cout) of dz [z > 0] and dz [z > 0] xhat -> crb_bn_relu_backward_partials(bn_y, dz, ...):
// that BatchNorm's backward launches no reduction pass over (bn_y, dz)
extern "C" int crb_conv3x3_winograd2_bnbwd_nhwc(const float* x, const float* U, float* y, float* stats, int N, int H, int W, int cin,
                                                int cout, const float* bn_y, const float* mean, const float* invstd, const float* gamma,
                                                const float* beta, int relu, void* stream) {
  if (!stats || !bn_y || !mean || !invstd || !gamma || !beta) return CRB_ERR_ARG;
  const Wino2Bnb b{bn_y, mean, invstd, gamma, beta, relu ? 1 : 0};
  return winograd2_launch(x, U, y, N, H, W, cin, cout, nullptr, 0, stream, stats, &b);
}
#endif

extern "C" int crb_conv3x3_winograd2_stats_nhwc(const float* x, const float* U, float* y, float* stats, int N, int H, int W, int cin,
                                                int cout, void* stream) {
  if (!stats) return CRB_ERR_ARG;
  return winograd2_launch(x, U, y, N, H, W, cin, cout, nullptr, 0, stream, stats);
}

// cus > 0: a kernel that will hold `cus` CUs for milliseconds is about to be launched on `stream` (call right before it, same
// stream); cus = 0: it has finished (call right after it, same stream). Persistent launches on other streams (the Winograd forward
// kernel) then spread their work over the CUs that are left instead of queueing a workgroup behind every taken one.
extern "C" int crb_cu_reservation(int cus, void* stream) {
  if (cus < 0) return CRB_ERR_ARG;
  int dev = 0;
  const int n_cu = device_cus(&dev);
  if (n_cu > 0 && cus > n_cu / 2) cus = n_cu / 2;             // at most half of the device is announced as taken
  hipLaunchKernelGGL(cu_busy_set_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, cus);
  CRB_CHECK_LAUNCH();
  return crbhip_wino4_cu_busy_set(cus, (hipStream_t)stream);
}
