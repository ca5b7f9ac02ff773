// Weight gradient of the 3x3 stride-1 pad-1 convolution on channels_last (NHWC) maps in the Winograd F(2x2, 3x3) domain on the
// f32 MFMA (row a7 of SURVEY §8: backward of pcdet/models/backbones_2d/base_bev_backbone.py:24-41; MIOpen's f32 implicit-GEMM
// wrw kernels took 16.2 ms of a 51 ms SECOND step after the forward / input-gradient convolutions moved to winograd_conv2.hip).
//
//   forward:   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A          per 2x2 output tile
//   therefore: dU[xi][ci][co] = sum_tiles V[xi][tile][ci] * M[xi][tile][co],   V = B^T d B (the forward's input transform),
//              M = A dY A^T (4x4 from the tile's 2x2 output gradient),   dg[ci][co] = G^T dU[.][ci][co] G   (3x3)
// i.e. 16 GEMMs (Cin x tiles) x (tiles x Cout): 2.25x fewer multiplications than the direct weight gradient, and BOTH operands
// are produced by transforms.
//
// Workgroup = 512 threads = 8 waves (two per SIMD, 128 accumulators each, the staged MFMA loop of winograd_conv2.hip) =
// one (64 input channels x 64 output channels) block of all 16 dU[xi] over a contiguous RANGE of tile chunks. The reduction
// index k runs over tiles: a chunk = 8 tiles = 2 tile rows x 4 tile columns of one image. Per chunk:
//   * the raw input block (6 x 10 pixels x 64 channels = 15 KB) comes in by LDS-DMA (global_load_lds_dwordx4), double buffered;
//     wave w transforms tile w of the chunk for its lane's channel: 16 ds_read_b32 (consecutive channels: conflict-free),
//     32 VALU, 8 ds_write_b64 into the V image;
//   * the output-gradient block of the chunk (4 x 8 pixels x 64 channels = 8 KB) comes in by LDS-DMA as well; M is NOT staged as
//     an image: it is the MFMA's A operand (rows = output channels), a lane needs M of ONE channel and two tiles per k-step pair,
//     i.e. 8 gradient values per chunk, and every M[xi] is a sum of at most four of them with coefficients +-1: the lanes read
//     their 8 values (ds_read_b32) and form the operands in registers, ~40 VALU per chunk (the first version transformed into a
//     second 32 KB image: 8 ds_write_b64 + 8 ds_read_b128 per lane and chunk more, and no LDS left for a third raw buffer);
//   * V image: [xi pair][k pair][row 0..63][k parity][xi parity] floats: a transform thread writes (xi even, xi odd) of its
//     (row, k) as one ds_write_b64 (lanes = consecutive rows: 2-way conflict, no more), an MFMA lane reads two k-steps of two xi
//     with one conflict-free ds_read_b128; V is the B operand (columns = input channels): a lane ends with dU of 4 consecutive
//     output channels for one input channel: 16-byte stores;
//   * both raw blocks are TRIPLE buffered (2 x 32 KB V + 3 x 16 KB + 3 x 8 KB = 136 KB): the DMA of a chunk goes out two barriers
//     before its data is needed and the wait before a barrier is vmcnt(3) - three DMA instructions per thread and chunk, loads
//     return in order - so the youngest chunk stays in flight across the barrier (every chunk's maps are first-touch HBM lines:
//     with one barrier of lead the waves were parked for them, 174 of 864 us).
// Every workgroup writes its partial dU block (256 KB); crb_winograd2_wgrad's second kernel adds the partials of a block in
// range order in double and applies G^T . G: bit-reproducible.
#include <type_traits>
#include <atomic>
#include "crb_common.h"
#include "../../include/crb_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BLK = 64;                       // channels per block on either side
constexpr int KT = 8;                         // tiles per chunk (2 tile rows x 4 tile columns)
constexpr int IMG_FLOATS = 16 * BLK * KT;     // 8192 = 32 KB
// raw input block: pixel rows 4 p - 1 .. 4 p + 4 (6), pixel columns 8 bc - 1 .. 8 bc + 8 (10)
constexpr int RAWX_FLOATS = 4096;             // 60 pixels x 64 channels = 960 DMA slots of 16 bytes, + 64 junk slots (wave 7's second)
constexpr int RAWG_FLOATS = 4 * 8 * BLK;      // 32 pixels x 64 channels = 512 slots = 8 KB
constexpr int LDS_FLOATS = 2 * IMG_FLOATS + 3 * RAWX_FLOATS + 3 * RAWG_FLOATS;
constexpr int NT = 512;

__device__ float g_wgrad_zero_page[64];       // source of out-of-map pixels / gradients (zero-initialised, never written)

// operand image of one chunk: float index of (xi, row, k)
__host__ __device__ __forceinline__ constexpr int imgb_index(int xi, int row, int k) {
  return (xi >> 1) * (BLK * 16) + (k >> 1) * (BLK * 4) + row * 4 + ((k & 1) << 1) + (xi & 1);
}

struct WgradArgs {
  const float* x;      // (N,H,W,Cin)
  const float* dy;     // (N,H,W,Cout)
  float* part;         // (ranges, nci * nco blocks, 16, 64 ci, 64 co)
  const float* affine; // AFFINE instances: (Cin, 2) per input channel (scale, shift): the layer's input is relu(scale * x + shift)
  const float* zero;   // g_wgrad_zero_page (as an argument: every use of the symbol itself costs a scalar load + wait)
  int N, H, W, cin, cout;
  int th, tw;          // tiles per column / row
  int tw4;             // chunk columns per image = ceil(tw / 4)
  int rp;              // chunk rows per image = ceil(th / 2)
  int nchunks;         // N * rp * tw4
  int nci, nco;        // channel blocks
  int nranges;         // K ranges (multiple of 8)
};

struct ChunkPos { int n, p, bc; };            // image, tile-row pair, tile-column block
__device__ __forceinline__ void chunk_next(ChunkPos& c, const WgradArgs& a) {
  if (++c.bc < a.tw4) return;
  c.bc = 0;
  if (++c.p < a.rp) return;
  c.p = 0;
  ++c.n;
}

__device__ __forceinline__ void glds16(const float* gsrc, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// MODE (measurement builds): 1 = no MFMAs, 2 = no transforms, 3 = no DMA in the loop (wrong results); 4 = correct results + cycle
// accounting per wave in g_wgrad_dbg (8 uint64 per wave): {total, stages 0-6, parked at wait + barrier, last stage, chunks}
__device__ unsigned long long* g_wgrad_dbg = nullptr;
// AFFINE: the convolution's input was relu(scale[c] * x + shift[c]) (applied by the forward kernel's input transform, never stored):
// the V transform applies the same to the values it reads, times the 0 / 1 mask of the patch positions inside the map (scalars:
// a wave transforms one tile)
template <int MODE, bool AFFINE = false>
__global__ __launch_bounds__(NT, 2) void winograd2_wgrad_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const Vb = lds;                       // V images (B operand), two buffers
  float* const Xb = lds + 2 * IMG_FLOATS;      // raw input blocks, three buffers
  float* const Gb = Xb + 3 * RAWX_FLOATS;      // raw output-gradient blocks, three buffers
  const int T = threadIdx.x, lane = T & 63, wave = __builtin_amdgcn_readfirstlane(T >> 6);       // (a scalar for the compiler)

  // ---- block and range of this workgroup: the nci * nco blocks of ONE range read the same maps: same XCD (id % 8)
  const int nblk = a.nci * a.nco;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int blk = slot % nblk, range = (slot / nblk) * 8 + xcd;
  if (range >= a.nranges) return;
  const int cib = blk / a.nco, cob = blk - cib * a.nco;
  const int c_first = (int)((int64_t)range * a.nchunks / a.nranges);
  const int c_end = (int)((int64_t)(range + 1) * a.nchunks / a.nranges);
  const int total = c_end - c_first;
  float* const out = a.part + ((int64_t)range * nblk + blk) * (16 * BLK * BLK);

  // ---- MFMA role: wave = 16 output channels (wk: A operand rows) x 32 input channels (wt: B operand, two 16-column blocks)
  const int wt = wave >> 2, wk = wave & 3;
  const int l15 = lane & 15, kq = lane >> 4;
  const int b_off = kq * (BLK * 4) + (wt * 32 + l15) * 4;
  // the lane's 8 gradient values: tiles 2 kq, 2 kq + 1 = tile row kq >> 1, tile columns 2 (kq & 1), + 1 of the chunk:
  // pixel rows 2 (kq >> 1) + {0, 1}, pixel columns 4 (kq & 1) + 0..3 of the 4 x 8 pixel block, channel wk * 16 + l15
  const int g_off = ((2 * (kq >> 1)) * 8 + 4 * (kq & 1)) * BLK + wk * 16 + l15;
  f32x4 acc[16][2];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) {
    acc[xi][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[xi][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  if (total <= 0) {                            // more ranges than chunks: a zero partial
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
      for (int cbk = 0; cbk < 2; ++cbk)
        *reinterpret_cast<f32x4*>(out + (xi * BLK + wt * 32 + cbk * 16 + l15) * BLK + wk * 16 + 4 * kq) = acc[xi][cbk];
    return;
  }
  ChunkPos first;
  {
    const int rows = c_first / a.tw4;
    first.bc = c_first - rows * a.tw4;
    first.n = rows / a.rp;
    first.p = rows - first.n * a.rp;
  }

  // ---- V transform role: wave w = tile w of the chunk (tile row w >> 2, tile column w & 3), lane = input channel of the block
  const int t_tr = wave >> 2, t_tc = wave & 3;
  const int raw_off = (2 * t_tr * 10 + 2 * t_tc) * BLK + lane;       // pixel (i, j) of the patch: + (i * 10 + j) * 64
  const int img_off = imgb_index(0, lane, wave);

  // ---- DMA: three 16-byte slots per thread and chunk. Input: 960 slots = (pixel 0..59, channel quad): slots T and T + 512 (the
  //      last 64, wave 7's, copy the zero page into the junk tail of the buffer: EVERY wave issues exactly three instructions, the
  //      count the vmcnt(3) below relies on). Gradient: 512 slots = (pixel 0..31 of the 4 x 8 block, channel quad): slot T.
  int s_r[2], s_c[2], s_q[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int q = T + NT * s, px = q >> 4;
    s_q[s] = q & 15;
    s_r[s] = px / 10;                          // >= 6 for the junk slots
    s_c[s] = px - s_r[s] * 10;
  }
  const int gs_r = (T >> 4) >> 3, gs_c = (T >> 4) & 7, gs_q = T & 15;
  const float* const xblk = a.x + cib * BLK;
  const float* const dyblk = a.dy + cob * BLK;
  ChunkPos dpos = first;                       // next chunk of the DMA
  int d_buf = 0;                               // its buffer (chunk index mod 3)
  // source addresses = chunk origin (scalar) + a per-thread constant, as 32-bit element offsets (the host bounds the maps);
  // prep_dma forms the three addresses of the next chunk among the MFMAs of stage 5, issue_dma behind the barrier is three
  // instructions (measured: with the addresses formed behind the barrier - 64-bit multiplies under divergent branches and a
  // scalar load of the zero page's address per slot - the last stage took 1,330 cycles instead of the 512 of its MFMAs)
  int lo_x[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) lo_x[s] = ((s_r[s] - 1) * a.W + (s_c[s] - 1)) * a.cin + s_q[s] * 4;
  const int lo_g = (gs_r * a.W + gs_c) * a.cout + gs_q * 4;
  const float* psrc[3];
  auto pick = [&](bool ok, const float* in_map) {      // ok ? in_map : zero page, as mask arithmetic: no divergent branches
    const uint64_t m = ok ? ~0ULL : 0ULL;
    return reinterpret_cast<const float*>((reinterpret_cast<uint64_t>(in_map) & m) | (reinterpret_cast<uint64_t>(a.zero) & ~m));
  };
  auto prep_dma = [&]() {
    const bool img = dpos.n < a.N;
    const int y0 = 4 * dpos.p, x0 = 8 * dpos.bc;
    const int origin = (dpos.n * a.H + y0) * a.W + x0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int y = y0 - 1 + s_r[s], xx = x0 - 1 + s_c[s];
      const bool ok = img & (s_r[s] < 6) & ((unsigned)y < (unsigned)a.H) & ((unsigned)xx < (unsigned)a.W);
      psrc[s] = pick(ok, xblk + (origin * a.cin + lo_x[s]));
    }
    // output pixel (4 p + r, 8 bc + c): rows / columns past the map (odd sizes, partial chunks) read zeros
    psrc[2] = pick(img & (y0 + gs_r < a.H) & (x0 + gs_c < a.W), dyblk + (origin * a.cout + lo_g));
    chunk_next(dpos, a);
  };
  auto issue_dma = [&]() {
    glds16(psrc[0], Xb + d_buf * RAWX_FLOATS + (wave * 64) * 4);
    glds16(psrc[1], Xb + d_buf * RAWX_FLOATS + (NT + wave * 64) * 4);
    glds16(psrc[2], Gb + d_buf * RAWG_FLOATS + wave * 64 * 4);
    d_buf = d_buf == 2 ? 0 : d_buf + 1;
  };
  auto wait_older_and_barrier = [&]() {        // everything but the three youngest vector-memory operations (the last chunk issued)
    asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  // ---- V transform
  // packed f32 (as the forward kernel's transform): a register pair = columns (2 h, 2 h + 1) of a patch row, 8 + 8 v_pk_add_f32
  // instead of 32 adds and the moves that packed the store operands
  f32x2 dp[4][2], tp[4][2];
  ChunkPos vpos = first;                       // AFFINE: chunk the transform works on (one ahead of the MFMAs)
  f32x2 sc2 = (f32x2){1.f, 1.f}, sh2 = (f32x2){0.f, 0.f};
  if (AFFINE) {
    const f32x2 sb = *reinterpret_cast<const f32x2*>(a.affine + (cib * BLK + lane) * 2);
    sc2 = (f32x2){sb[0], sb[0]};
    sh2 = (f32x2){sb[1], sb[1]};
  }
  auto v_load = [&](const float* raw) {
    const float* p = raw + raw_off;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) dp[i][h] = (f32x2){p[(i * 10 + 2 * h) * BLK], p[(i * 10 + 2 * h + 1) * BLK]};
  };
  // AFFINE: activation of the column pair h, in the stage that consumes it (placed behind the reads in stage 0 it made the wave
  // wait for them there, in front of that stage's MFMAs: +80 us per call). Patch positions outside the map are zero padding of
  // the ACTIVATED map: 0 / 1 factors (scalars: a wave transforms one tile; branch-free - a branch splits the stage's basic block)
  float cm[4], rm[4];
  auto v_act_setup = [&]() {
    const int y0 = 4 * vpos.p - 1 + 2 * t_tr, x0 = 8 * vpos.bc - 1 + 2 * t_tc;
#pragma unroll
    for (int j = 0; j < 4; ++j) cm[j] = (x0 + j >= 0 && x0 + j < a.W) ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) rm[i] = (y0 + i >= 0 && y0 + i < a.H) ? 1.f : 0.f;
    chunk_next(vpos, a);
  };
  auto v_act = [&](int h) {
    const f32x2 zero = (f32x2){0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
      dp[i][h] = __builtin_elementwise_max(__builtin_elementwise_fma(dp[i][h], sc2, sh2), zero) * (f32x2){rm[i] * cm[2 * h], rm[i] * cm[2 * h + 1]};
  };
  auto v_cols = [&](int h) {        // B^T d, B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1], columns 2 h and 2 h + 1
    if (AFFINE) {
      if (h == 0) v_act_setup();
      v_act(h);
    }
    tp[0][h] = dp[0][h] - dp[2][h];
    tp[1][h] = dp[1][h] + dp[2][h];
    tp[2][h] = dp[2][h] - dp[1][h];
    tp[3][h] = dp[1][h] - dp[3][h];
  };
  auto v_row = [&](float* V, int i) {          // xi = 4 i .. 4 i + 3 as two (even, odd) pairs: (t0 - t2, t1 + t2), (t2 - t1, t1 - t3)
    f32x2 o01, o23;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(o01) : "v"(tp[i][0]), "v"(tp[i][1]));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(o23) : "v"(tp[i][0]), "v"(tp[i][1]));
    *reinterpret_cast<f32x2*>(V + img_off + imgb_index(i * 4 + 0, 0, 0)) = o01;
    *reinterpret_cast<f32x2*>(V + img_off + imgb_index(i * 4 + 2, 0, 0)) = o23;
  };

  // ---- M = A dY A^T in registers, packed f32 and with the two negations of A folded into the accumulators' final sign.
  //      gp[e][r] = gradient of tile 2 kq + e at pixel row r, columns (0, 1). A = [1 0; 1 1; 1 -1; 0 -1]:
  //      (A dY)[i] = (g0, g0 + g1, g0 - g1, -g1)[i] -> mip[e][i] = (g0, g0 + g1, g0 - g1, +g1): row 3 carries the opposite sign;
  //      M[i][j] = (a, a + b, a - b, -b)[j] with (a, b) = (A dY)[i] -> (mip[i][0], ms[i][0], ms[i][1], +mip[i][1]): column 3
  //      carries the opposite sign. Accumulator xi = 4 i + j therefore holds sign(i) sign(j) dU, sign(3) = -1: xi = 3, 7, 11, 12,
  //      13, 14 are negated when the partial is stored. 12 packed instructions per chunk, none between the MFMAs (it was 48
  //      scalar ones, most of them feeding the next MFMA directly).
  f32x2 gp[2][2], mip[2][4], ms[2][4];
  auto g_load = [&](const float* G) {
    const float* p = G + g_off;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int r = 0; r < 2; ++r) gp[e][r] = (f32x2){p[(r * 8 + 2 * e) * BLK], p[(r * 8 + 2 * e + 1) * BLK]};
  };
  auto m_cols = [&]() {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      mip[e][0] = gp[e][0];
      mip[e][1] = gp[e][0] + gp[e][1];
      mip[e][2] = gp[e][0] - gp[e][1];
      mip[e][3] = gp[e][1];
#pragma unroll
      for (int i = 0; i < 4; ++i)               // (a + b, a - b) from (a, b)
        asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(ms[e][i]) : "v"(mip[e][i]));
    }
  };
  auto m_val = [&](int e, int xi) {
    const int i = xi >> 2, j = xi & 3;
    return j == 0 ? mip[e][i][0] : j == 1 ? ms[e][i][0] : j == 2 ? ms[e][i][1] : mip[e][i][1];
  };
  auto acc_sign = [](int xi) { return (((xi >> 2) == 3) != ((xi & 3) == 3)) ? -1.f : 1.f; };

  // ---- MFMA pieces (as winograd_conv2.hip: 8 stages = xi pairs, the B operands of the next pair read one stage ahead)
  f32x4 v0[2], v1[2];
  auto op_read = [&](const float* V, int xp, int slot) {
    v0[slot] = *reinterpret_cast<const f32x4*>(V + xp * (BLK * 16) + b_off);
    v1[slot] = *reinterpret_cast<const f32x4*>(V + xp * (BLK * 16) + b_off + 16 * 4);
  };
  // the 8 MFMAs of an xi pair in k-step-major order: the two MFMAs of one accumulator are four instructions apart (two apart,
  // a wave that has the matrix pipe to itself waits for the first one's result: measured +0.5% here, +2.6% in the forward kernel)
  auto mfma_pair = [&](int xp, int slot) {     // register 2 e + h of a V read: k-step e, xi parity h
    float m[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) { m[h][0] = m_val(0, 2 * xp + h); m[h][1] = m_val(1, 2 * xp + h); }
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int xi = 2 * xp + h;
        if (MODE == 1) {
          acc[xi][0][0] += m[h][e] * v0[slot][2 * e + h];
          acc[xi][1][0] += m[h][e] * v1[slot][2 * e + h];
        } else {
          acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(m[h][e], v0[slot][2 * e + h], acc[xi][0], 0, 0, 0);
          acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(m[h][e], v1[slot][2 * e + h], acc[xi][1], 0, 0, 0);
        }
      }
  };
  // chunk g (buffers: V image g & 1, raw blocks g mod 3 -> gbuf / next ones): MFMAs on V(g) and the lane's M(g); V transform of
  // chunk g + 1 from raw input g + 1 (stage 0), the lane's gradient values of chunk g + 1 (stage 6); at the barrier the raw blocks
  // of chunk g + 2 have landed (issued two barriers ago) and chunk g + 4's DMA goes out into the buffers of chunk g + 1
  int gbuf = 0;                                // raw buffer of chunk g
  unsigned long long cyc[4] = {0, 0, 0, 0};
  const unsigned long long cyc_start = MODE == 4 ? __builtin_amdgcn_s_memtime() : 0ULL;
  auto chunk = [&](int g, auto do_issue, auto do_t) {
    const int cur = g & 1, nxt = cur ^ 1;
    const int gb1 = gbuf == 2 ? 0 : gbuf + 1;  // raw buffer of chunk g + 1
    const float* V = Vb + cur * IMG_FLOATS;
    float* Vn = Vb + nxt * IMG_FLOATS;
    constexpr bool T_ON = decltype(do_t)::value && MODE != 2;
    unsigned long long c0 = 0, c1 = 0;
    if (MODE == 4) c0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int xp = 0; xp < 8; ++xp) {
      if (T_ON) {
        if (xp == 0) v_load(Xb + gb1 * RAWX_FLOATS);
        if (xp == 1) v_cols(0);
        if (xp == 2) v_cols(1);
        if (xp >= 3 && xp < 7) v_row(Vn, xp - 3);
      }
      // the lane's gradient values of chunk g + 1 (landed since the previous barrier), BEFORE this chunk's barrier: behind it
      // the DMA of chunk g + 4 overwrites both raw buffers of chunk g + 1
      if (xp == 6 && decltype(do_t)::value) g_load(Gb + gb1 * RAWG_FLOATS);
      if (xp == 5 && MODE != 3 && decltype(do_issue)::value) prep_dma();
      if (xp < 7) op_read(V, xp + 1, (xp + 1) & 1);
      if (xp == 7) {
        if (MODE == 4) { c1 = __builtin_amdgcn_s_memtime(); cyc[1] += c1 - c0; }
        if (MODE == 4) {
          asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
          const unsigned long long cw = __builtin_amdgcn_s_memtime();
          cyc[0] += cw - c1;
        }
        wait_older_and_barrier();
        if (MODE == 4) { c0 = __builtin_amdgcn_s_memtime(); cyc[2] += c0 - c1; }
        if (MODE != 3 && decltype(do_issue)::value) issue_dma();
      }
      mfma_pair(xp, xp & 1);
      if (xp == 7 && decltype(do_t)::value) {
        op_read(Vn, 0, 0);
        if (MODE != 2) m_cols();                // (the MFMAs above have read their operands)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 4) cyc[3] += __builtin_amdgcn_s_memtime() - c0;
    gbuf = gb1;
  };
  using std::true_type;
  using std::false_type;

  // ---- prologue: chunks 0, 1 land; chunk 0 -> V(0) and the lane's M(0) while chunks 2, 3 go out
  prep_dma();
  issue_dma();
  prep_dma();
  issue_dma();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  prep_dma();
  issue_dma();                                   // chunk 2 -> buffer 2
  g_load(Gb);
  m_cols();
  v_load(Xb);
  v_cols(0);
  v_cols(1);
#pragma unroll
  for (int i = 0; i < 4; ++i) v_row(Vb, i);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  prep_dma();
  issue_dma();                                   // chunk 3 -> buffer 0 (chunk 0's raw blocks are consumed)
  op_read(Vb, 0, 0);

  // chunk g's barrier issues chunk g + 4 (past the range: another range's chunk or zeros, never used - the count stays 3)
  int g = 0;
  for (; g + 1 < total; ++g) chunk(g, true_type{}, true_type{});
  chunk(g, false_type{}, false_type{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // DMA must not outlive the workgroup

  // ---- partial dU: lane = input channel l15 of the wave's block, 4 consecutive output channels
#pragma unroll
  for (int xi = 0; xi < 16; ++xi)
#pragma unroll
    for (int cbk = 0; cbk < 2; ++cbk)
      *reinterpret_cast<f32x4*>(out + (xi * BLK + wt * 32 + cbk * 16 + l15) * BLK + wk * 16 + 4 * kq) = acc[xi][cbk] * acc_sign(xi);
  if (MODE == 4 && g_wgrad_dbg && lane == 0) {
    unsigned long long* o = g_wgrad_dbg + ((int64_t)blockIdx.x * 8 + wave) * 8;
    o[0] = __builtin_amdgcn_s_memtime() - cyc_start;
    o[1] = cyc[1]; o[2] = cyc[2]; o[3] = cyc[3]; o[4] = (unsigned long long)total; o[5] = cyc[0];
  }
}

// dW[co][ci][ky][kx] = (G^T dU[.][ci][co] G)[ky][kx], dU = sum over the ranges in range order (double), written with the element
// strides of the weight tensor. G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]
__global__ __launch_bounds__(256) void winograd2_wgrad_reduce_kernel(const float* __restrict__ part, int nranges, int nci, int nco,
                                                                     float* __restrict__ dw, int64_t so, int64_t si, int64_t sky,
                                                                     int64_t skx, int cin, int cout) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)cin * cout) return;
  const int ci = (int)(t / cout), co = (int)(t - (int64_t)ci * cout);
  const int blk = (ci / BLK) * nco + co / BLK;
  const int64_t nblk = (int64_t)nci * nco;
  const float* p = part + blk * (int64_t)(16 * BLK * BLK) + (ci % BLK) * BLK + co % BLK;
  double u[16];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) u[xi] = 0.0;
  for (int r = 0; r < nranges; ++r) {
    const float* q = p + r * nblk * (16 * BLK * BLK);
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) u[xi] += (double)q[xi * BLK * BLK];
  }
  double h[3][4];                  // G^T dU: rows of dU combined
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[0][j] = u[0 * 4 + j] + 0.5 * (u[1 * 4 + j] + u[2 * 4 + j]);
    h[1][j] = 0.5 * (u[1 * 4 + j] - u[2 * 4 + j]);
    h[2][j] = 0.5 * (u[1 * 4 + j] + u[2 * 4 + j]) + u[3 * 4 + j];
  }
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    float* o = dw + co * so + ci * si + ky * sky;
    o[0 * skx] = (float)(h[ky][0] + 0.5 * (h[ky][1] + h[ky][2]));
    o[1 * skx] = (float)(0.5 * (h[ky][1] - h[ky][2]));
    o[2 * skx] = (float)(0.5 * (h[ky][1] + h[ky][2]) + h[ky][3]);
  }
}

__host__ int wgrad_ranges(int nblk) {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 8;
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  int r = (n_cu / nblk) & ~7;
  return r < 8 ? 8 : r;
}

}  // namespace

CRB_KNOB g_wgrad2_mode = 0;     // measurement builds: 1 = no MFMAs, 2 = no transforms, 3 = no DMA in the loop
#ifdef CRB_MEASURE
extern "C" int crb_winograd2_wgrad_set_mode(int mode) { g_wgrad2_mode = (mode >= 1 && mode <= 4) ? mode : 0; return CRB_OK; }
extern "C" int crb_winograd2_wgrad_set_debug(void* dev_buf) {       // mode 4: 8 waves x 8 uint64 per workgroup, NULL = off
  unsigned long long* p = (unsigned long long*)dev_buf;
  CRB_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wgrad_dbg), &p, sizeof(p)));
  return CRB_OK;
}
#endif

extern "C" int crb_winograd2_wgrad_supported(int cin, int cout, int H, int W) {
  return (cin > 0 && cout > 0 && cin % BLK == 0 && cout % BLK == 0 && H >= 1 && W >= 1) ? 1 : 0;
}

extern "C" int64_t crb_winograd2_wgrad_workspace_bytes(int cin, int cout) {
  if (!crb_winograd2_wgrad_supported(cin, cout, 1, 1)) return 0;
  const int nblk = (cin / BLK) * (cout / BLK);
  return (int64_t)wgrad_ranges(nblk) * nblk * 16 * BLK * BLK * 4;
}

static int winograd2_wgrad_launch(const float* x, const float* affine, const float* dy, float* dw, int64_t so, int64_t si,
                                  int64_t sky, int64_t skx, int N, int H, int W, int cin, int cout, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
  if (N <= 0 || H <= 0 || W <= 0) return CRB_ERR_ARG;
  if (!crb_winograd2_wgrad_supported(cin, cout, H, W)) return CRB_ERR_UNSUPPORTED;
  if (workspace_bytes < crb_winograd2_wgrad_workspace_bytes(cin, cout) || !workspace) return CRB_ERR_WORKSPACE;
  if (((int64_t)N * H + 8) * (W + 16) * (cin > cout ? cin : cout) >= (1LL << 31)) return CRB_ERR_ARG;   // 32-bit element offsets
  // per-device state (a process may drive several devices, from several threads): the zero page's address and the "dynamic LDS
  // attribute set" flags belong to the device the call runs on
  static std::atomic<const float*> zero_pages[64];
  static std::atomic<unsigned> attr_done[64];
  int dev = 0;
  CRB_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return CRB_ERR_ARG;
  const float* zero_page = zero_pages[dev].load(std::memory_order_acquire);
  if (!zero_page) {
    CRB_HIP(hipGetSymbolAddress((void**)&zero_page, HIP_SYMBOL(g_wgrad_zero_page)));
    zero_pages[dev].store(zero_page, std::memory_order_release);
  }
  WgradArgs a;
  a.x = x; a.dy = dy; a.part = (float*)workspace; a.zero = zero_page; a.affine = affine;
  a.N = N; a.H = H; a.W = W; a.cin = cin; a.cout = cout;
  a.th = (H + 1) / 2; a.tw = (W + 1) / 2;
  a.tw4 = (a.tw + 3) / 4;
  a.rp = (a.th + 1) / 2;
  const int64_t nch = (int64_t)N * a.rp * a.tw4;
  if (nch >= (1LL << 30)) return CRB_ERR_ARG;
  a.nchunks = (int)nch;
  a.nci = cin / BLK; a.nco = cout / BLK;
  const int nblk = a.nci * a.nco;
  a.nranges = wgrad_ranges(nblk);
  const size_t lds = LDS_FLOATS * sizeof(float);
  auto kern = winograd2_wgrad_kernel<0>;
#ifdef CRB_MEASURE
  if (g_wgrad2_mode == 1) kern = winograd2_wgrad_kernel<1>;
  if (g_wgrad2_mode == 2) kern = winograd2_wgrad_kernel<2>;
  if (g_wgrad2_mode == 3) kern = winograd2_wgrad_kernel<3>;
  if (g_wgrad2_mode == 4) kern = winograd2_wgrad_kernel<4>;
#endif
  const unsigned bit = 1u << (g_wgrad2_mode & 7);
  if (!(attr_done[dev].load(std::memory_order_acquire) & bit)) {
    CRB_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done[dev].fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.nranges * nblk)), dim3(NT), lds, (hipStream_t)stream, a);
  CRB_CHECK_LAUNCH();
  const int64_t per = (int64_t)cin * cout;
  hipLaunchKernelGGL(winograd2_wgrad_reduce_kernel, dim3(crb_cdiv(per, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, a.nranges, a.nci, a.nco, dw, so, si, sky, skx, cin, cout);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

// x (N,H,W,Cin), dy (N,H,W,Cout) f32 NHWC -> dw = gradient of the nn.Conv2d weight (Cout,Cin,3,3), written with the element
// strides (so, si, sky, skx) of that tensor. workspace: crb_winograd2_wgrad_workspace_bytes(cin, cout).
extern "C" int crb_winograd2_wgrad(const float* x, const float* dy, float* dw, int64_t so, int64_t si, int64_t sky, int64_t skx,
                                   int N, int H, int W, int cin, int cout, void* workspace, int64_t workspace_bytes, void* stream) {
  return winograd2_wgrad_launch(x, nullptr, dy, dw, so, si, sky, skx, N, H, W, cin, cout, workspace, workspace_bytes, stream);
}
