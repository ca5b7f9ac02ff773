// 3x3 stride-1 pad-1 convolution on channels_last (NHWC) f32 maps as Winograd F(2x2, 3x3) with the 16 GEMMs on the bf16 matrix
// pipe through an EXACT three-way split of every f32 operand (round 6; row a7 of SURVEY §8: the BEV backbone's 3x3 convolutions,
// pcdet/models/backbones_2d/base_bev_backbone.py:24-41).
//
// Why: on gfx950 the exact-f32 MFMA (v_mfma_f32_16x16x4_f32 / 32x32x2) runs at the f32 VECTOR rate, 1/16 of the bf16 matrix rate.
// winograd_conv2.hip sits at 0.65 of that roof for two rounds with every ingredient priced at 2-5 %. An f32
// value is the exact sum of three bf16 values
//     x = x1 + x2 + x3,  x1 = x & 0xffff0000, x2 = (x - x1) & 0xffff0000, x3 = x - x1 - x2      (both subtractions exact, x3 has
//                                                                                               <= 8 significant bits: no rounding)
// and a product of two such sums has nine exact partial products of which the six with i + j <= 4 carry everything above
// 2^-24 |x w| (the three dropped ones are bounded by 2^-16 * 2^-8 * 2 + 2^-32 < 2^-23 |x w|, the size of ONE f32 rounding of the
// product): six v_mfma_f32_32x32x16_bf16 passes, f32 accumulation in the matrix pipe, give the f32 GEMM at 16 / 6 = 2.67 x the
// f32 MFMA rate. Measured against f64 convolutions the results are as close as the f32-MFMA kernel's (tests/test_winograd_gpu.py,
// tests/test_insitu_gpu.py: same bars). The weights are split once per step by the weight-image kernel; V = B^T d B is split by the
// lanes that form it (and / subtract / v_perm: no rounding instruction), so the MFMA waves read ready bf16 fragments.
//
// With the matrix pipe 2.67 x faster the kernel is bound by LDS traffic and VALU issue, and the design follows from that:
//   * workgroup = 256 threads = ONE wave per SIMD with up to 512 registers: wave = 32 tiles x 32 output channels x all 16 xi
//     (256 accumulators), workgroup = 64 tiles (16 tile rows x 4 tile columns of one spatial block, tile rows running over the
//     whole batch) x 64 output channels. All xi of a tile live in one lane: Y = A^T M A happens in registers, 16-byte stores.
//   * chunk = 16 input channels = the K of one MFMA; a chunk runs as FOUR phases, one xi row (4 xi) each: per phase and wave 24
//     MFMAs (768 matrix-pipe cycles) on 12 A + 12 B fragments (ds_read_b128). LDS holds two V phase images (26 KB each, written by
//     the transform), three U phase images (24 KB each, LDS-DMA two phases ahead) and ONE raw block (24 KB) = 147 KB.
//   * the raw block is shared by the whole workgroup: 36 pixel rows x 10 pixels x 16 channels (halo rows fetched once; a block
//     that straddles two images keeps a 2-row gap between them), DMA'd once per chunk right after its last reader and read in ONE
//     phase: thread (tile, channel quad) reads its 4 x 4 patch (16 ds_read_b128, conflict-free through an even/odd pixel-column
//     order of the LDS image), does the column pass t = B^T d and keeps t (64 registers) for the four row passes of the next
//     four phases.
//   * one barrier per phase: [wait own V stores + the DMA that has to have landed | barrier | DMA of raw(c+1) / U(f+2) | row pass +
//     split + stores of V(f+1) interleaved with the 24 MFMAs of phase f].
//   * persistent workgroups (one per CU), contiguous unit ranges, busy-CU latch as in winograd_conv2.hip.
#include <atomic>
#include <type_traits>
#include "crb_common.h"
#include "../../include/crb_hip.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TB_ROWS = 16, TB_COLS = 4;     // tile block: 16 tile rows x 4 tile columns
constexpr int WG_K = 64;                     // output channels per workgroup
constexpr int CC = 16;                       // input channels per chunk
constexpr int NT = 256;
// byte sizes of the LDS images
constexpr int U_XP = 2 * 64 * 16;            // one (xi, piece): [k group 2][row 64][8 bf16] = 2048
constexpr int U_PHASE = 12 * U_XP;           // 4 xi x 3 pieces = 24576
constexpr int V_REGION = 64 * 16 + 64;       // [tile 64][8 bf16] + 64 bytes: the two k groups of a b64 store land in different banks
constexpr int V_XP = 2 * V_REGION;           // 2176
constexpr int V_PHASE = 12 * V_XP;           // 26112
constexpr int RAW_ROWS = 36, RAW_PX = 10;
constexpr int RAW_SLOTS = 6 * NT;            // 1536 slots of 16 bytes (1440 used: 36 rows x 10 pixels x 4 channel quads)
constexpr int RAW_BYTES = RAW_SLOTS * 16;    // 24576
constexpr int LDS_V = 0, LDS_U = 2 * V_PHASE, LDS_RAW = LDS_U + 3 * U_PHASE, LDS_BYTES [[maybe_unused]] = LDS_RAW + RAW_BYTES;   // 150528
constexpr int LDS_RAW_R = 2 * V_PHASE, LDS_BYTES_R = LDS_RAW_R + RAW_BYTES;                                     // 76800 (U in registers)

__device__ float g_wino4_zero_page[64];      // source of out-of-map pixels (zero-initialised, never written)
__device__ int g_cu_busy4 = 0;               // see winograd_conv2.hip (crb_cu_reservation sets both)
__device__ unsigned g_cu_latch4[64];
#ifdef CRB_MEASURE
__device__ unsigned long long* g_wino4_dbg = nullptr;      // measurement mode 9: 8 uint64 per (workgroup, wave)
#endif
__global__ void cu_busy4_set_kernel(int v) { __hip_atomic_store(&g_cu_busy4, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// exact three-way split: the bf16 bit patterns (high halves) of x1, x2, x3
__device__ __forceinline__ void split3(float x, unsigned& h1, unsigned& h2, unsigned& h3) {
  const unsigned b = __float_as_uint(x);
  const float r1 = x - __uint_as_float(b & 0xffff0000u);
  const unsigned b1 = __float_as_uint(r1);
  const float r2 = r1 - __uint_as_float(b1 & 0xffff0000u);
  h1 = b >> 16;
  h2 = b1 >> 16;
  h3 = __float_as_uint(r2) >> 16;
}

// ---- weight image. Byte offset of (xi = 4 i + jx, piece p, kernel input channel ci, kernel output channel co):
//      [co / 64][ci / 16][i][jx][p][k group (ci % 16) / 8][row co % 64][element ci % 8] bf16  - a phase image is one linear 24 KB copy
__device__ __forceinline__ void weights_octet(const float* __restrict__ w, int64_t so, int64_t si, int64_t sky, int64_t skx,
                                              unsigned char* __restrict__ U, int kin, int kout, int mode_, int64_t t) {
  if (t >= (int64_t)(kin / 8) * kout) return;
  const int mode = mode_ & 1;                  // 0 = forward weights, 1 = input-gradient weights (transposed, rotated)
  const bool layout_c = (mode_ & 2) != 0;      // image layout of the 32 x 128 form
  const int cg = (int)(t / kout), co = (int)(t - (int64_t)cg * kout);     // channel octet, kernel-side output channel
  u32x4 out[16][3];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ci = cg * 8 + e;
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
        g[a][b] = mode == 0 ? w[co * so + ci * si + a * sky + b * skx] : w[ci * so + co * si + (2 - a) * sky + (2 - b) * skx];
    float tmp[4][3];                 // G g (the expressions of winograd_conv2.hip: the f32 U is the same number)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      tmp[0][b] = g[0][b];
      tmp[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
      tmp[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
      tmp[3][b] = g[2][b];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float u[4] = {tmp[r][0], 0.5f * (tmp[r][0] + tmp[r][1] + tmp[r][2]), 0.5f * (tmp[r][0] - tmp[r][1] + tmp[r][2]), tmp[r][2]};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        unsigned h[3];
        split3(u[c], h[0], h[1], h[2]);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const unsigned d = out[r * 4 + c][p][e >> 1];
          out[r * 4 + c][p][e >> 1] = (e & 1) ? (d | (h[p] << 16)) : h[p];
        }
      }
    }
  }
  const int nch = kin / CC;
  const int chunk = cg >> 1, kg = cg & 1;
  if (layout_c) {      // [co / 128][ci / 16][i][jx][k group][row co % 128][p][element ci % 8]: the image of winograd4c_kernel
    const int cb = co / 128, row = co - cb * 128;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const int64_t off = ((int64_t)(cb * nch + chunk) * 4 + (xi >> 2)) * (4 * 2 * 128 * 48) + (xi & 3) * (2 * 128 * 48) + (kg * 128 + row) * 48 + p * 16;
        *reinterpret_cast<u32x4*>(U + off) = out[xi][p];
      }
    return;
  }
  const int cb = co / WG_K, row = co - cb * WG_K;
#pragma unroll
  for (int xi = 0; xi < 16; ++xi)
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const int64_t off = ((int64_t)(cb * nch + chunk) * 4 + (xi >> 2)) * U_PHASE + (((xi & 3) * 3 + p) * 2 + kg) * (64 * 16) + row * 16;
      *reinterpret_cast<u32x4*>(U + off) = out[xi][p];
    }
}

__global__ __launch_bounds__(256) void winograd4_weights_conv_kernel(const float* __restrict__ w, int64_t so, int64_t si, int64_t sky,
                                                                     int64_t skx, unsigned char* __restrict__ U, int kin, int kout,
                                                                     int mode) {
  weights_octet(w, so, si, sky, skx, U, kin, kout, mode, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

constexpr int WJ_MAX = 32;
struct Wino4WJob { const float* w; unsigned char* U; int64_t so, si, sky, skx; int kin, kout, mode, first_block; };
struct Wino4WJobs { int n; Wino4WJob job[WJ_MAX]; };

__global__ __launch_bounds__(256) void winograd4_weights_conv_multi_kernel(Wino4WJobs jobs) {
  int j = 0;
  while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.job[j + 1].first_block) ++j;       // (wave-uniform, <= 31 steps)
  const Wino4WJob& q = jobs.job[j];
  weights_octet(q.w, q.so, q.si, q.sky, q.skx, q.U, q.kin, q.kout, q.mode, (int64_t)(blockIdx.x - q.first_block) * 256 + threadIdx.x);
}

struct Wino4Args {
  const float* x;            // (N,H,W,Cin)
  const unsigned char* U;    // crb_winograd4_weights_conv image
  float* y;                  // (N,H,W,Cout)
  const float* bias;         // (Cout) or null
  float* stats;              // null, or (2 * spatial blocks, 2, Cout): per (spatial block, half of its 64 tiles) the column sums of y and
                             // y^2 over the outputs inside the map (crb_bn_relu_forward_partials takes them)
  int N, H, W, cin, cout, relu;
  int th, tw;                // tile rows per image = ceil(H / 2), tiles per row = ceil(W / 2)
  int RT;                    // tile rows over the batch = N * th
  int tw4;                   // tile-column blocks = ceil(tw / 4)
  int nblocks;               // spatial blocks = ceil(RT / 16) * tw4
  int ncb;                   // cout / 64
  unsigned seq;              // launch sequence number for the busy-CU latch; 0 = ignore g_cu_busy4
};

template <int AUX = 0>
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, AUX);
}

// 16 bytes per lane from a uniform base + a lane offset, issued HERE and waited for by hand (s_waitcnt vmcnt(n) before the first use:
// the compiler does not know this is a memory operation - with LDS-DMA copies in flight beside ordinary loads its own bookkeeping
// falls back to vmcnt(0), which would wait for the copy that was just requested)
template <int OFF>
__device__ __forceinline__ bf16x8 gload16(const unsigned char* sbase, unsigned voff) {
  bf16x8 r;
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(r) : "v"(voff), "s"(sbase), "n"(OFF));
  return r;
}

__device__ __forceinline__ f32x4 sload4(const float* p) {
  f32x4 r;
  asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
  return r;
}

struct UnitPos {
  int cb, bc, R0, n0, ty0;     // channel block, tile-column block, first tile row over the batch = image n0, row ty0 of it
};
__device__ __forceinline__ UnitPos unit_at(int u, const Wino4Args& a) {       // units are numbered with the channel block fastest
  UnitPos p;
  const int tb = u / a.ncb;
  p.cb = u - tb * a.ncb;
  const int br = tb / a.tw4;
  p.bc = tb - br * a.tw4;
  p.R0 = br * TB_ROWS;
  p.n0 = p.R0 / a.th;
  p.ty0 = p.R0 - p.n0 * a.th;
  return p;
}

// The 256 accumulators are the AGPRs a0 .. a255 BY NAME (xi -> a[16 xi : 16 xi + 15]): as C++ values the register allocator
// parks some of the sixteen 512-bit tuples in VGPRs and copies them to AGPRs around every MFMA (first build of this file: 240 +
// 268 copies and 455 spilled registers in the loop). The compiler sees these registers through the clobber list at the head of
// the kernel only, so nothing here may spill (it would spill into "free" AGPRs): the build checks for zero scratch.
// Hazards the compiler cannot see inside asm: MFMA -> MFMA on the same accumulator needs no wait states (srcC = vDst, same
// opcode); MFMA -> v_accvgpr_read needs the MFMA's passes + wait states: acc_settle() in front of the output transform.
template <int XI>
__device__ __forceinline__ void mfma_acc(const bf16x8& A, const bf16x8& B) {
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%2:%3], %0, %1, a[%2:%3]" : : "v"(A), "v"(B), "n"(XI * 16), "n"(XI * 16 + 15));
}
template <int R>
__device__ __forceinline__ float acc_read() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(R));
  return x;
}
template <int R>
__device__ __forceinline__ void acc_zero() {
  asm volatile("v_accvgpr_write_b32 a[%0], 0" : : "n"(R));
}
template <int R0, int N>
__device__ __forceinline__ void acc_zero_range() {
  if constexpr (N > 0) {
    acc_zero<R0>();
    acc_zero_range<R0 + 1, N - 1>();
  }
}
__device__ __forceinline__ void acc_settle() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

// MODE (measurement builds, wrong results): 1 = no MFMAs, 2 = no transform (V never written), 3 = no DMA after the prologue, 4 = no
// operand reads, 5 = no U copies, 6 = no raw copies, 7 = no V stores (values formed), 8 = no epilogue stores
// UR = 1: the U fragments go from L2 straight to the registers of the MFMA lanes (global_load_dwordx4, one phase ahead, into the
// registers the previous phase's MFMAs of the same xi just released) instead of LDS-DMA + ds_read: through LDS the A fragments
// cost 24 KB of copies landing + 48 KB read back per phase (half of the operand reads) on a copy engine that tops out at 45-60
// B/clk per CU (profiles/r06_probe_lds_dma_stream_rate.txt). LDS: two V images + the raw block = 76.8 KB.
// Tried on top of this form and not faster (all bit-equal; commit 1c6ae13, profiles/r06_time_winograd4_v5_experiments_not_faster.txt):
// fragments two phases ahead (row pass formed per phase from two raw rows to pay for the registers, two raw blocks); V formed two
// phases ahead with the next phase's first B fragments read before the barrier; finished units leaving through an LDS staging block
// as full 256-byte rows during the next unit; workgroups staggered against synchronised store bursts (-5 us for a 27 us delay);
// ordinary (compiler-tracked) fragment loads with the raw copies hidden in inline asm (counted lgkmcnt waits, 3 % slower).
template <int MODE, int UR>
__global__ __launch_bounds__(NT, 1) void winograd4_kernel(Wino4Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const Vb = lds + LDS_V;
  unsigned char* const Ub = lds + LDS_U;
  unsigned char* const Rb = lds + (UR ? LDS_RAW_R : LDS_RAW);
  const int T = threadIdx.x, lane = T & 63, wave = __builtin_amdgcn_readfirstlane(T >> 6);
  asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17",
               "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35",
               "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53",
               "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71",
               "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89",
               "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106",
               "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121",
               "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136",
               "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151",
               "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166",
               "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181",
               "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196",
               "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211",
               "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226",
               "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241",
               "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");

  // ---- the unit range of this workgroup
  const int nunits = a.nblocks * a.ncb;
  int G = gridDim.x;
  if (a.seq) {
    if (T == 0) {
      unsigned* L = g_cu_latch4 + (a.seq & 63u);
      unsigned v = __hip_atomic_load(L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), mine;
      for (;;) {
        if ((v >> 8) == a.seq) { mine = v & 255u; break; }
        const int b = __hip_atomic_load(&g_cu_busy4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned busy = (unsigned)min(max(b, 0), 255);
        const unsigned seen = atomicCAS(L, v, (a.seq << 8) | busy);
        if (seen == v) { mine = busy; break; }
        v = seen;
      }
      *reinterpret_cast<unsigned*>(Rb) = mine;
    }
    __syncthreads();
    const int busy = (int)*reinterpret_cast<const unsigned*>(Rb);
    __syncthreads();
    G = max(1, (int)gridDim.x - __builtin_amdgcn_readfirstlane(busy));
    if ((int)blockIdx.x >= G) return;
  }
  const int u_first = (int)((int64_t)blockIdx.x * nunits / G);
  const int u_end = (int)((int64_t)(blockIdx.x + 1) * nunits / G);
  if (u_first >= u_end) return;
  const int nch = a.cin / CC;
  const int total_chunks = (u_end - u_first) * nch;

  // ---- MFMA role: wave = tile half (wave & 1) x channel half (wave >> 1)
  const int w_th = wave & 1, w_kh = wave >> 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int a_rd = lhi * (64 * 16) + (w_kh * 32 + l31) * 16;       // U image: A operand, rows = output channels
  const int b_rd = lhi * V_REGION + (w_th * 32 + l31) * 16;        // V image: B operand, columns = tiles

  // ---- transform role: thread = (tile, channel quad)
  const int t_tile = T >> 2, t_q = T & 3, t_tr = t_tile >> 2, t_tc = t_tile & 3;
  // V store: [xi][piece][k group = quad >> 1][tile][half = quad & 1] -> byte offset inside a phase image (+ (jx * 3 + p) * V_XP)
  const int v_wr = (t_q >> 1) * V_REGION + t_tile * 16 + (t_q & 1) * 8;
  int t_unit = u_first, tcnt = 0;                   // unit and chunk of it that the NEXT t_load reads
  // raw read base of the thread's patch: local pixel row 2 tr (+ 2 behind an image boundary), pixel column order (even | odd)
  auto raw_base = [&](int u) {
    const UnitPos p = unit_at(u, a);
    const int rows_a = min(TB_ROWS, a.th - p.ty0);           // tile rows of the block that belong to its first image
    const int lr = 2 * t_tr + (t_tr >= rows_a ? 2 : 0);
    return 16 * (4 * (lr * RAW_PX + t_tc) + t_q);
  };
  int raw_rd = raw_base(t_unit);

  // ---- DMA role. raw: slots s = T + 256 j (16 bytes each): s -> (pixel position P = s >> 2, channel quad s & 3), P = local
  //      row * 10 + column order (x >> 1) + 5 (x & 1). U: 6 x 16 bytes per thread and phase.
  int r_unit = u_first, rc = 0;                     // unit / chunk of it the next raw copy fetches
  const float* rsrc[6];
  unsigned rstep = 0;                               // bit j: slot j moves on by a chunk per issue (in-map pixel)
  auto raw_sources = [&](int u) {
    const UnitPos p = unit_at(u, a);
    const int rows_a = min(TB_ROWS, a.th - p.ty0);
    const int limit_a = 2 * rows_a + 2;
    rstep = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int s = T + NT * j;
      const int P = s >> 2, q = s & 3;
      const int lr = P / RAW_PX, rem = P - lr * RAW_PX;
      const int xx = rem < 5 ? 2 * rem : 2 * (rem - 5) + 1;
      const bool in_a = lr < limit_a;
      const int n = in_a ? p.n0 : p.n0 + 1;
      const int py = in_a ? 2 * p.ty0 - 1 + lr : lr - limit_a - 1;
      const int px = 8 * p.bc - 1 + xx;
      const bool ok = lr < RAW_ROWS && (in_a || (rows_a < TB_ROWS && (lr - limit_a) < 2 * (TB_ROWS - rows_a) + 2)) && n < a.N &&
                      py >= 0 && py < a.H && px >= 0 && px < a.W;
      const int64_t off = (((int64_t)n * a.H + py) * a.W + px) * a.cin + q * 4;
      rsrc[j] = ok ? a.x + off : g_wino4_zero_page;
      rstep |= ok ? (1u << j) : 0u;
    }
  };
  raw_sources(r_unit);
  auto issue_raw = [&]() {
#pragma unroll
    for (int j = 0; j < 6; ++j) glds16(rsrc[j], Rb + (NT * j + wave * 64) * 16);
  };
  auto r_advance = [&]() {                                    // after issue_raw: move the sources to the next chunk
    if (++rc < nch) {
#pragma unroll
      for (int j = 0; j < 6; ++j) rsrc[j] += (rstep >> j & 1u) ? CC : 0;
      return;
    }
    rc = 0;
    ++r_unit;
    if (r_unit % a.ncb == 0) raw_sources(r_unit);            // next spatial block (channel block 0 again)
    else {
#pragma unroll
      for (int j = 0; j < 6; ++j) rsrc[j] -= (rstep >> j & 1u) ? (nch - 1) * CC : 0;
    }
  };
  int u_cb = u_first % a.ncb, upc = 0;              // channel block / phase of its unit the next U copy fetches
  const int u_lane = UR ? 0 : T * 16;               // (UR: usrc stays uniform, the lane part is a_rd in the load's offset register)
  const unsigned char* usrc = a.U + (int64_t)u_cb * nch * 4 * U_PHASE + u_lane;
  auto issue_u = [&](int ub) {                                // one phase image -> U buffer at byte offset ub
#pragma unroll
    for (int k = 0; k < 6; ++k) glds16(usrc + k * (NT * 16), Ub + ub + (k * NT + wave * 64) * 16);
  };
  auto u_advance = [&]() {
    if (++upc < nch * 4) { usrc += U_PHASE; return; }
    upc = 0;
    if (++u_cb == a.ncb) u_cb = 0;
    usrc = a.U + (int64_t)u_cb * nch * 4 * U_PHASE + u_lane;
  };
  bf16x8 Ar[4][3];                                   // UR: the A fragments of the phase about to run
  auto a_load = [&](auto jc) __attribute__((always_inline)) {    // fragments of xi jx of the image at usrc
    constexpr int jx = decltype(jc)::value;
    if (MODE == 5 || MODE == 3) {
      asm volatile("" : "=v"(Ar[jx][0]), "=v"(Ar[jx][1]), "=v"(Ar[jx][2]));
      return;
    }
    const unsigned voff = (unsigned)a_rd + (jx * 3 + 1) * U_XP;
    Ar[jx][0] = gload16<-U_XP>(usrc, voff);
    Ar[jx][1] = gload16<0>(usrc, voff);
    Ar[jx][2] = gload16<U_XP>(usrc, voff);
  };

  acc_zero_range<0, 256>();

  // ---- output transform of a finished unit: Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1]. Lane = tile l31 of the wave's half, channels
  //      cb * 64 + 32 kh + 8 j + 4 lhi + (0..3) for register group j (accumulator register 4 j + e of every xi)
  int e_unit = u_first, ec = 0;
  auto epilogue_group = [&](auto jc, const UnitPos& eu, float* yo, bool in, bool x1, bool y1) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    const int kbase = eu.cb * WG_K + w_kh * 32 + 4 * lhi + 8 * j;
    f32x4 bias = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
      const float* bp = a.bias + eu.cb * WG_K + __builtin_amdgcn_readfirstlane(w_kh) * 32 + 8 * j;
      const f32x4 b0 = sload4(bp), b1 = sload4(bp + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) bias[e] = lhi ? b1[e] : b0[e];
    }
    f32x4 M[16];
#define CRB_ACC4(XI) M[XI] = (f32x4){acc_read<XI * 16 + 4 * j>(), acc_read<XI * 16 + 4 * j + 1>(), acc_read<XI * 16 + 4 * j + 2>(), acc_read<XI * 16 + 4 * j + 3>()}
    CRB_ACC4(0); CRB_ACC4(1); CRB_ACC4(2); CRB_ACC4(3); CRB_ACC4(4); CRB_ACC4(5); CRB_ACC4(6); CRB_ACC4(7);
    CRB_ACC4(8); CRB_ACC4(9); CRB_ACC4(10); CRB_ACC4(11); CRB_ACC4(12); CRB_ACC4(13); CRB_ACC4(14); CRB_ACC4(15);
#undef CRB_ACC4
    f32x4 t0[4], t1[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      t0[s] = M[0 * 4 + s] + M[1 * 4 + s] + M[2 * 4 + s];
      t1[s] = M[1 * 4 + s] - M[2 * 4 + s] - M[3 * 4 + s];
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
    f32x4 s1 = (f32x4){0.f, 0.f, 0.f, 0.f}, s2 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (in && MODE != 8) {
      float* o = yo + 8 * j;
      *reinterpret_cast<f32x4*>(o) = y00;
      if (x1) *reinterpret_cast<f32x4*>(o + a.cout) = y01;
      if (y1) *reinterpret_cast<f32x4*>(o + (int64_t)a.W * a.cout) = y10;
      if (x1 && y1) *reinterpret_cast<f32x4*>(o + (int64_t)a.W * a.cout + a.cout) = y11;
      if (a.stats) {                                          // fixed order: (0,0), (0,1), (1,0), (1,1)
        const float m01 = x1 ? 1.f : 0.f, m10 = y1 ? 1.f : 0.f, m11 = (x1 && y1) ? 1.f : 0.f;
        s1 = y00; s2 = y00 * y00;
        s1 = s1 + y01 * m01; s2 = s2 + (y01 * y01) * m01;
        s1 = s1 + y10 * m10; s2 = s2 + (y10 * y10) * m10;
        s1 = s1 + y11 * m11; s2 = s2 + (y11 * y11) * m11;
      }
    }
    if (a.stats) {
      // sum over the 32 tiles of the lane half (rotations inside the 16-lane rows, then the two rows of the half: fixed order), lanes
      // 0 and 32 write their four channels: slab = (spatial block, tile half)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float u = s1[e], v = s2[e];
#define CRB_ROW_ROR_ADD(x, ctrl) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, false))
        CRB_ROW_ROR_ADD(u, 0x128); CRB_ROW_ROR_ADD(v, 0x128);       // row_ror:8
        CRB_ROW_ROR_ADD(u, 0x124); CRB_ROW_ROR_ADD(v, 0x124);
        CRB_ROW_ROR_ADD(u, 0x122); CRB_ROW_ROR_ADD(v, 0x122);
        CRB_ROW_ROR_ADD(u, 0x121); CRB_ROW_ROR_ADD(v, 0x121);
#undef CRB_ROW_ROR_ADD
        u += __shfl_xor(u, 16, 64);
        v += __shfl_xor(v, 16, 64);
        s1[e] = u;
        s2[e] = v;
      }
      if (l31 == 0) {
        const int64_t blk = (int64_t)(eu.R0 / TB_ROWS) * a.tw4 + eu.bc;
        float* so = a.stats + ((blk * 2 + w_th) * 2) * a.cout + kbase;
        *reinterpret_cast<f32x4*>(so) = s1;
        *reinterpret_cast<f32x4*>(so + a.cout) = s2;
      }
    }
  };
  auto unit_epilogue = [&]() __attribute__((always_inline)) {
    const UnitPos eu = unit_at(e_unit, a);
    const int tile = w_th * 32 + l31;
    const int tr = tile >> 2, tc = tile & 3;
    const int Rg = eu.R0 + tr;
    const int tx = eu.bc * TB_COLS + tc;
    int n2 = eu.n0, ty2 = eu.ty0 + tr;
    while (ty2 >= a.th) { ty2 -= a.th; ++n2; }
    const int oy = 2 * ty2, ox = 2 * tx;
    const bool in = Rg < a.RT && tx < a.tw;
    const bool x1 = ox + 1 < a.W, y1 = oy + 1 < a.H;
    float* const yo = a.y + (((int64_t)n2 * a.H + oy) * a.W + ox) * a.cout + eu.cb * WG_K + w_kh * 32 + 4 * lhi;
    acc_settle();
    epilogue_group(std::integral_constant<int, 0>{}, eu, yo, in, x1, y1);
    epilogue_group(std::integral_constant<int, 1>{}, eu, yo, in, x1, y1);
    epilogue_group(std::integral_constant<int, 2>{}, eu, yo, in, x1, y1);
    epilogue_group(std::integral_constant<int, 3>{}, eu, yo, in, x1, y1);
    acc_zero_range<0, 256>();
    ++e_unit;
  };

  // ---- input transform, cut into pieces that the phase places one behind each MFMA (one wave per SIMD: what is issued between
  //      two MFMAs runs under the first one; MFMAs back to back make the wave wait for the matrix pipe):
  //      col<b>: column b of the thread's 4 x 4 patch of 4 channels from the raw block, tp[.][b] = B^T d,
  //              B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1];
  //      xi<r, jx, s>: element jx of row r of tp B in five steps: value + first piece, second piece, third piece, pack + store of
  //              piece 0, pack + store of pieces 1 and 2 (3 ds_write_b64)
  f32x4 tp[4][4];
  f32x4 sv, sr1, sr2;                                 // value, first and second remainder of the element being split
  auto t_read = [&](f32x4 (&d)[4][4]) {
    const unsigned char* p = Rb + raw_rd;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int b = 0; b < 4; ++b)          // pixel column b of the patch: position t_tc + (b >> 1) + 5 (b & 1) of the row
        d[i][b] = *reinterpret_cast<const f32x4*>(p + (i * RAW_PX + (b >> 1) + 5 * (b & 1)) * 64);
  };
  auto t_col = [&](const f32x4 (&d)[4][4], int b) __attribute__((always_inline)) {
    tp[0][b] = d[0][b] - d[2][b];
    tp[1][b] = d[1][b] + d[2][b];
    tp[2][b] = d[2][b] - d[1][b];
    tp[3][b] = d[1][b] - d[3][b];
    // (anchors: volatile asm statements keep their order, so the values exist HERE, between the two MFMAs around this piece - the
    // optimizer otherwise sinks pure arithmetic to its first use, across sched_barrier)
    asm volatile("" : "+v"(tp[0][b]), "+v"(tp[1][b]), "+v"(tp[2][b]), "+v"(tp[3][b]));
  };
  auto trunc16 = [](float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); };
  auto pack_hi = [](const f32x4& x) {
    u32x2 o;
    o[0] = (__float_as_uint(x[0]) >> 16) | (__float_as_uint(x[1]) & 0xffff0000u);
    o[1] = (__float_as_uint(x[2]) >> 16) | (__float_as_uint(x[3]) & 0xffff0000u);
    return o;
  };
  auto t_xi = [&](unsigned char* V, auto rcst, auto jcst, auto scst) __attribute__((always_inline)) {
    constexpr int r = decltype(rcst)::value, jx = decltype(jcst)::value, s = decltype(scst)::value;
    if (MODE == 2) return;
    if constexpr (s == 0) {
      if constexpr (jx == 0) sv = tp[r][0] - tp[r][2];
      else if constexpr (jx == 1) sv = tp[r][1] + tp[r][2];
      else if constexpr (jx == 2) sv = tp[r][2] - tp[r][1];
      else sv = tp[r][1] - tp[r][3];
      asm volatile("" : "+v"(sv));
    } else if constexpr (s == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) sr1[e] = sv[e] - trunc16(sv[e]);
      asm volatile("" : "+v"(sr1));
    } else if constexpr (s == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) sr2[e] = sr1[e] - trunc16(sr1[e]);
      asm volatile("" : "+v"(sr2));
    } else if constexpr (s == 3) {
      if (MODE == 7) { asm volatile("" :: "v"(pack_hi(sv))); return; }          // measurement: no V stores
      *reinterpret_cast<u32x2*>(V + v_wr + (jx * 3 + 0) * V_XP) = pack_hi(sv);
    } else {
      if (MODE == 7) { asm volatile("" :: "v"(pack_hi(sr1)), "v"(pack_hi(sr2))); return; }
      *reinterpret_cast<u32x2*>(V + v_wr + (jx * 3 + 1) * V_XP) = pack_hi(sr1);
      *reinterpret_cast<u32x2*>(V + v_wr + (jx * 3 + 2) * V_XP) = pack_hi(sr2);
    }
  };
  auto t_advance = [&]() {                           // after a t_read: the next one reads the next chunk's block
    if (++tcnt < nch) return;
    tcnt = 0;
    ++t_unit;
    if (t_unit % a.ncb == 0) raw_rd = raw_base(t_unit);
  };

  // ---- phases. f = 4 chunk + i: MFMAs of xi row i on V[f & 1], U[f % 3]; V(f + 1) formed meanwhile; U(f + 2) and, in phase 0,
  //      raw(chunk + 1) requested right behind the barrier. Every phase requests its copies unconditionally (past the end of the
  //      workgroup's range they fetch a valid U image and zero-page / in-map pixels that nobody reads): the counts below hold in
  //      every phase.
  int ub_cur = 0, ub_nxt = U_PHASE, ub_nn = 2 * U_PHASE;
  auto phase = [&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    constexpr int rn = (i + 1) & 3;
    // DMA that may stay in flight: what phase f - 1 requested (U(f + 1); in phase 1 also raw(chunk + 1), requested before it)
    if (UR) {     // (register loads are waited for where they are used; raw(chunk + 1) of phase 0 has to be in LDS for phase 3: only
                  // what phase 2 requested may be in flight)
      if (i == 3) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (i == 1) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    unsigned char* const Vn = Vb + ((i + 1) & 1) * V_PHASE;
    const unsigned char* const Vc = Vb + (i & 1) * V_PHASE + b_rd;
    const unsigned char* const Uc = Ub + ub_cur + a_rd;
    bf16x8 A[2][3], B[2][3];
    auto op_read = [&](int jx, int slot) __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        if (MODE == 4) {                    // measurement: no operand reads
          asm volatile("" : "=v"(A[slot][p]), "=v"(B[slot][p]));
          continue;
        }
        if (!UR) A[slot][p] = *reinterpret_cast<const bf16x8*>(Uc + (jx * 3 + p) * U_XP);
        B[slot][p] = *reinterpret_cast<const bf16x8*>(Vc + (jx * 3 + p) * V_XP);
      }
    };
    op_read(0, 0);
    f32x4 d[4][4];
    if (i == 3) { t_read(d); t_advance(); }
    if (MODE != 3) {
      if (i == 0 && MODE != 6) { issue_raw(); if (!UR) r_advance(); }
      if (MODE != 5 && !UR) { issue_u(ub_nn); u_advance(); }
    }
    __builtin_amdgcn_sched_barrier(0);
    // piece k (behind MFMA k of the phase). Phase 3 starts with the four column passes (columns 0, 2 first: xi 0 needs them)
    auto piece = [&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      constexpr int k0 = (i == 3) ? k - 4 : k;
      if constexpr (i == 3 && k < 4) {
        t_col(d, k == 0 ? 0 : k == 1 ? 2 : k == 2 ? 1 : 3);
      } else if constexpr (k0 >= 0 && k0 < 20) {
        t_xi(Vn, std::integral_constant<int, rn>{}, std::integral_constant<int, k0 / 5>{}, std::integral_constant<int, k0 % 5>{});
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto stage = [&](auto jc) __attribute__((always_inline)) {
      constexpr int jx = decltype(jc)::value;
      constexpr int s = jx & 1;
      const bf16x8 &A0 = UR ? Ar[jx][0] : A[s][0], &A1 = UR ? Ar[jx][1] : A[s][1], &A2 = UR ? Ar[jx][2] : A[s][2];
      // register loads in flight behind the fragments of (this phase, xi jx): the other xi of the previous phase's requests, this
      // phase's requests so far, and in phase 0 the six raw copies
      if (UR) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(i == 0 ? 15 : 9) : "memory");
      if (MODE != 1) mfma_acc<4 * i + jx>(A0, B[s][2]);
      if (jx < 3) op_read(jx + 1, s ^ 1);
      piece(std::integral_constant<int, 6 * jx + 0>{});
      if (MODE != 1) mfma_acc<4 * i + jx>(A2, B[s][0]);
      piece(std::integral_constant<int, 6 * jx + 1>{});
      if (MODE != 1) mfma_acc<4 * i + jx>(A1, B[s][1]);
      piece(std::integral_constant<int, 6 * jx + 2>{});
      if (MODE != 1) mfma_acc<4 * i + jx>(A0, B[s][1]);
      piece(std::integral_constant<int, 6 * jx + 3>{});
      if (MODE != 1) mfma_acc<4 * i + jx>(A1, B[s][0]);
      piece(std::integral_constant<int, 6 * jx + 4>{});
      if (MODE != 1) mfma_acc<4 * i + jx>(A0, B[s][0]);
      if (UR) {                                  // the next phase's fragments of this xi, into the registers just released
        if (MODE == 1) asm volatile("" :: "v"(Ar[jx][0]), "v"(Ar[jx][1]), "v"(Ar[jx][2]));
        a_load(jc);
        if (jx == 3) u_advance();
      }
      piece(std::integral_constant<int, 6 * jx + 5>{});
    };
    stage(std::integral_constant<int, 0>{});
    stage(std::integral_constant<int, 1>{});
    stage(std::integral_constant<int, 2>{});
    stage(std::integral_constant<int, 3>{});
    const int t = ub_cur; ub_cur = ub_nxt; ub_nxt = ub_nn; ub_nn = t;
    if (UR && i == 0 && MODE != 3 && MODE != 6) r_advance();      // (its branches behind the MFMAs: the compiler's vmcnt bookkeeping
                                                                  // of the register loads gives up at control-flow joins)
  };

  // ---- prologue: raw(0), U(0), U(1) in one round trip, raw(0) -> tp -> V(0)
  issue_raw(); r_advance();
  if (UR) {
    a_load(std::integral_constant<int, 0>{}); a_load(std::integral_constant<int, 1>{});
    a_load(std::integral_constant<int, 2>{}); a_load(std::integral_constant<int, 3>{});
    u_advance();
  } else {
    issue_u(0); u_advance();
    issue_u(U_PHASE); u_advance();
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
    f32x4 d[4][4];
    t_read(d); t_advance();
#pragma unroll
    for (int b = 0; b < 4; ++b) t_col(d, b);
  }
#define CRB_T_XI_ALL(JX)                                                                                                      \
  t_xi(Vb, std::integral_constant<int, 0>{}, std::integral_constant<int, JX>{}, std::integral_constant<int, 0>{});            \
  t_xi(Vb, std::integral_constant<int, 0>{}, std::integral_constant<int, JX>{}, std::integral_constant<int, 1>{});            \
  t_xi(Vb, std::integral_constant<int, 0>{}, std::integral_constant<int, JX>{}, std::integral_constant<int, 2>{});            \
  t_xi(Vb, std::integral_constant<int, 0>{}, std::integral_constant<int, JX>{}, std::integral_constant<int, 3>{});            \
  t_xi(Vb, std::integral_constant<int, 0>{}, std::integral_constant<int, JX>{}, std::integral_constant<int, 4>{})
  CRB_T_XI_ALL(0); CRB_T_XI_ALL(1); CRB_T_XI_ALL(2); CRB_T_XI_ALL(3);
#undef CRB_T_XI_ALL

  for (int cg = 0; cg < total_chunks; ++cg) {
    phase(std::integral_constant<int, 0>{});
    phase(std::integral_constant<int, 1>{});
    phase(std::integral_constant<int, 2>{});
    phase(std::integral_constant<int, 3>{});
    if (++ec == nch) { ec = 0; unit_epilogue(); }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (copies requested past the end of the range)
}


// ================================================================================================================================
// Third form (round 6, "c"): workgroup tile = 32 tiles (8 tile rows x 4 tile columns) x 128 output channels instead of 64 x 64.
// The product GEMM per phase is the same (four waves x one 32 x 32 block x 4 xi), but the input side halves: V of a spatial block is
// formed ONCE per 128 output channels instead of once per 64 (the row pass, its split, the V stores and the raw copies were three of
// the additive costs of the form above), every wave reads the same B fragments, and the four waves load four different channel
// quarters of the weight image (no two waves ask for the same fragment any more). All 256 threads still share the transform: thread =
// (tile, channel quad, xi pair), the xi pair {0,1} or {2,3} of a row - its three lane-local patch columns X0, X1, X2 are
// (c0, c2, c1) or (c2, c1, -c3), so that both halves run the same instructions: xi_a = X0 - X1, xi_b = X1 + X2 (exact: a negation
// and a commuted addition). Same arithmetic as the form above: the outputs are bit-equal.
// Weight image "c": [Cout/128][Cin/16][xi row][xi][k group][row 128][piece 3][8 bf16] - a lane's three pieces are 48 contiguous bytes.
namespace c4 {
constexpr int TB_ROWS = 8, TB_COLS = 4;
constexpr int WG_K = 128;
constexpr int U_ROW = 48;
constexpr int U_XI = 2 * WG_K * U_ROW;         // 12288
constexpr int U_PHASE = 4 * U_XI;              // 49152
constexpr int V_REGION = 32 * 16 + 64;         // [tile 32][8 bf16] + 64 bytes between the two k groups
constexpr int V_XP = 2 * V_REGION + 16;        // 1168: + 16 so that the two xi pairs of a (tile, quad) store to different banks
constexpr int V_PHASE = 12 * V_XP;             // 14016
constexpr int RAW_ROWS = 20, RAW_PX = 10;
constexpr int RAW_SLOTS = 4 * NT;              // 1024 slots of 16 bytes (800 used: 20 rows x 10 pixels x 4 channel quads)
constexpr int RAW_BYTES = RAW_SLOTS * 16;      // 16384
constexpr int LDS_V = 0, LDS_RAW = 4 * V_PHASE, LDS_BYTES = LDS_RAW + 2 * RAW_BYTES;   // 88832: four V rows (two buffers of two xi rows), two raw blocks

struct UnitPos { int cb, bc, R0, n0, ty0; };
__device__ __forceinline__ UnitPos unit_at(int u, const Wino4Args& a) {
  UnitPos p;
  const int tb = u / a.ncb;
  p.cb = u - tb * a.ncb;
  const int br = tb / a.tw4;
  p.bc = tb - br * a.tw4;
  p.R0 = br * TB_ROWS;
  p.n0 = p.R0 / a.th;
  p.ty0 = p.R0 - p.n0 * a.th;
  return p;
}
}  // namespace c4

__global__ __launch_bounds__(NT, 1) void winograd4c_kernel(Wino4Args a) {
  constexpr int TB_ROWS = c4::TB_ROWS, TB_COLS = c4::TB_COLS, WG_K = c4::WG_K, U_ROW = c4::U_ROW, U_XI = c4::U_XI, U_PHASE = c4::U_PHASE,
                V_REGION = c4::V_REGION, V_XP = c4::V_XP, V_PHASE = c4::V_PHASE, RAW_ROWS = c4::RAW_ROWS, RAW_PX = c4::RAW_PX;
  using UnitPos = c4::UnitPos;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const Vb = lds + c4::LDS_V;
  unsigned char* const Rb = lds + c4::LDS_RAW;
  const int T = threadIdx.x, lane = T & 63, wave = __builtin_amdgcn_readfirstlane(T >> 6);
  asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17",
               "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35",
               "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53",
               "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71",
               "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89",
               "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106",
               "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121",
               "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136",
               "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151",
               "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166",
               "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181",
               "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196",
               "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211",
               "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226",
               "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241",
               "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");

  // ---- the unit range of this workgroup (busy-CU latch as in the forms above)
  const int nunits = a.nblocks * a.ncb;
  int G = gridDim.x;
  if (a.seq) {
    if (T == 0) {
      unsigned* L = g_cu_latch4 + (a.seq & 63u);
      unsigned v = __hip_atomic_load(L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), mine;
      for (;;) {
        if ((v >> 8) == a.seq) { mine = v & 255u; break; }
        const int b = __hip_atomic_load(&g_cu_busy4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned busy = (unsigned)min(max(b, 0), 255);
        const unsigned seen = atomicCAS(L, v, (a.seq << 8) | busy);
        if (seen == v) { mine = busy; break; }
        v = seen;
      }
      *reinterpret_cast<unsigned*>(Rb) = mine;
    }
    __syncthreads();
    const int busy = (int)*reinterpret_cast<const unsigned*>(Rb);
    __syncthreads();
    G = max(1, (int)gridDim.x - __builtin_amdgcn_readfirstlane(busy));
    if ((int)blockIdx.x >= G) return;
  }
  const int u_first = (int)((int64_t)blockIdx.x * nunits / G);
  const int u_end = (int)((int64_t)(blockIdx.x + 1) * nunits / G);
  if (u_first >= u_end) return;
  const int nch = a.cin / CC;
  const int total_chunks = (u_end - u_first) * nch;

  // ---- MFMA role: wave = channel quarter, all 32 tiles
  const int l31 = lane & 31, lhi = lane >> 5;
  const int a_rd = (lhi * WG_K + wave * 32 + l31) * U_ROW;          // weight image: the lane's (k group, row), three pieces
  const int b_rd = lhi * V_REGION + l31 * 16;                       // V image: B operand, columns = tiles

  // ---- transform role: thread = (tile, channel quad, xi pair)
  const int t_tile = T >> 3, t_q = (T >> 1) & 3, t_h = T & 1, t_tr = t_tile >> 2, t_tc = t_tile & 3;
  const int v_wr = (t_q >> 1) * V_REGION + t_tile * 16 + (t_q & 1) * 8 + t_h * (6 * V_XP);
  const float t_sg = t_h ? -1.f : 1.f;                              // X2 = -c3 for the pair {2, 3}
  int t_unit = u_first, tcnt = 0;
  // raw read offsets of the lane's three patch columns: X0 = c0 | c2, X1 = c2 | c1, X2 = c1 | c3. Pixel x (0..9) of a block row sits at
  // position pos(x) = x with 1 <-> 2 and 7 <-> 8 swapped: the order in which none of the three ds_read_b128 of a row has two lanes of
  // a 16-lane group on one bank group with different addresses (found by enumeration for thread = (tile, quad, xi pair); the
  // even | odd order of the 64 x 64 form gives 2-way conflicts here: SQ_LDS_BANK_CONFLICT 6.8 M cycles per launch)
  auto px_pos = [](int x) { return (x == 1 || x == 2) ? 3 - x : (x == 7 || x == 8) ? 15 - x : x; };
  int raw_rd[3];
  auto raw_base = [&](int u) {
    const UnitPos p = c4::unit_at(u, a);
    const int rows_a = min(TB_ROWS, a.th - p.ty0);
    const int lr = 2 * t_tr + (t_tr >= rows_a ? 2 : 0);
    const int b0 = t_h ? 2 : 0, b1 = t_h ? 1 : 2, b2 = t_h ? 3 : 1;
    raw_rd[0] = 16 * (4 * (lr * RAW_PX + px_pos(2 * t_tc + b0)) + t_q);
    raw_rd[1] = 16 * (4 * (lr * RAW_PX + px_pos(2 * t_tc + b1)) + t_q);
    raw_rd[2] = 16 * (4 * (lr * RAW_PX + px_pos(2 * t_tc + b2)) + t_q);
  };
  raw_base(t_unit);

  // ---- copy role. raw: slots s = T + 256 j (16 bytes each): s -> (pixel position P = s >> 2, channel quad s & 3), P = local
  //      row * 10 + pos(x)
  int r_unit = u_first, rc = 0;
  const float* rsrc[4];
  unsigned rstep = 0;
  auto raw_sources = [&](int u) {
    const UnitPos p = c4::unit_at(u, a);
    const int rows_a = min(TB_ROWS, a.th - p.ty0);
    const int limit_a = 2 * rows_a + 2;
    rstep = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int s = T + NT * j;
      const int P = s >> 2, q = s & 3;
      const int lr = P / RAW_PX, rem = P - lr * RAW_PX;
      const int xx = px_pos(rem);                               // (pos is its own inverse)
      const bool in_a = lr < limit_a;
      const int n = in_a ? p.n0 : p.n0 + 1;
      const int py = in_a ? 2 * p.ty0 - 1 + lr : lr - limit_a - 1;
      const int px = 8 * p.bc - 1 + xx;
      const bool ok = lr < RAW_ROWS && (in_a || (rows_a < TB_ROWS && (lr - limit_a) < 2 * (TB_ROWS - rows_a) + 2)) && n < a.N &&
                      py >= 0 && py < a.H && px >= 0 && px < a.W;
      const int64_t off = (((int64_t)n * a.H + py) * a.W + px) * a.cin + q * 4;
      rsrc[j] = ok ? a.x + off : g_wino4_zero_page;
      rstep |= ok ? (1u << j) : 0u;
    }
  };
  raw_sources(r_unit);
  int r_par = 0, t_par = 0;                         // raw block (of two) the next copy fills / the next transform read takes
  auto issue_raw = [&]() {
    unsigned char* const dst = Rb + r_par * c4::RAW_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(rsrc[j], dst + (NT * j + wave * 64) * 16);
    r_par ^= 1;
  };
  auto r_advance = [&]() {
    if (++rc < nch) {
#pragma unroll
      for (int j = 0; j < 4; ++j) rsrc[j] += (rstep >> j & 1u) ? CC : 0;
      return;
    }
    rc = 0;
    ++r_unit;
    if (r_unit % a.ncb == 0) raw_sources(r_unit);
    else {
#pragma unroll
      for (int j = 0; j < 4; ++j) rsrc[j] -= (rstep >> j & 1u) ? (nch - 1) * CC : 0;
    }
  };
  int u_cb = u_first % a.ncb, upc = 0;              // channel block / phase of its unit the next fragment loads fetch
  const unsigned char* usrc = a.U + (int64_t)u_cb * nch * 4 * U_PHASE;
  auto u_advance = [&]() {
    if (++upc < nch * 4) { usrc += U_PHASE; return; }
    upc = 0;
    if (++u_cb == a.ncb) u_cb = 0;
    usrc = a.U + (int64_t)u_cb * nch * 4 * U_PHASE;
  };
  bf16x8 Ar[2][4][3];                                // the A fragments of the next two phases (set = phase parity)
  auto a_load = [&](auto sc, auto jc) __attribute__((always_inline)) {
    constexpr int st = decltype(sc)::value, jx = decltype(jc)::value;
    const unsigned voff = (unsigned)a_rd + jx * U_XI;
    Ar[st][jx][0] = gload16<0>(usrc, voff);
    Ar[st][jx][1] = gload16<16>(usrc, voff);
    Ar[st][jx][2] = gload16<32>(usrc, voff);
  };

  acc_zero_range<0, 256>();

  // ---- output transform of a finished unit: lane = tile l31, channels cb * 128 + 32 wave + 8 j + 4 lhi + (0..3) for register group j
  int e_unit = u_first, ec = 0;
  auto epilogue_group = [&](auto jc, const UnitPos& eu, float* yo, bool in, bool x1, bool y1) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    const int kbase = eu.cb * WG_K + wave * 32 + 4 * lhi + 8 * j;
    f32x4 bias = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
      const float* bp = a.bias + eu.cb * WG_K + wave * 32 + 8 * j;
      const f32x4 b0 = sload4(bp), b1 = sload4(bp + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) bias[e] = lhi ? b1[e] : b0[e];
    }
    f32x4 M[16];
#define CRB_ACC4(XI) M[XI] = (f32x4){acc_read<XI * 16 + 4 * j>(), acc_read<XI * 16 + 4 * j + 1>(), acc_read<XI * 16 + 4 * j + 2>(), acc_read<XI * 16 + 4 * j + 3>()}
    CRB_ACC4(0); CRB_ACC4(1); CRB_ACC4(2); CRB_ACC4(3); CRB_ACC4(4); CRB_ACC4(5); CRB_ACC4(6); CRB_ACC4(7);
    CRB_ACC4(8); CRB_ACC4(9); CRB_ACC4(10); CRB_ACC4(11); CRB_ACC4(12); CRB_ACC4(13); CRB_ACC4(14); CRB_ACC4(15);
#undef CRB_ACC4
    f32x4 t0[4], t1[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      t0[s] = M[0 * 4 + s] + M[1 * 4 + s] + M[2 * 4 + s];
      t1[s] = M[1 * 4 + s] - M[2 * 4 + s] - M[3 * 4 + s];
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
    f32x4 s1 = (f32x4){0.f, 0.f, 0.f, 0.f}, s2 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (in) {
      float* o = yo + 8 * j;
      *reinterpret_cast<f32x4*>(o) = y00;
      if (x1) *reinterpret_cast<f32x4*>(o + a.cout) = y01;
      if (y1) *reinterpret_cast<f32x4*>(o + (int64_t)a.W * a.cout) = y10;
      if (x1 && y1) *reinterpret_cast<f32x4*>(o + (int64_t)a.W * a.cout + a.cout) = y11;
      if (a.stats) {                                          // fixed order: (0,0), (0,1), (1,0), (1,1)
        const float m01 = x1 ? 1.f : 0.f, m10 = y1 ? 1.f : 0.f, m11 = (x1 && y1) ? 1.f : 0.f;
        s1 = y00; s2 = y00 * y00;
        s1 = s1 + y01 * m01; s2 = s2 + (y01 * y01) * m01;
        s1 = s1 + y10 * m10; s2 = s2 + (y10 * y10) * m10;
        s1 = s1 + y11 * m11; s2 = s2 + (y11 * y11) * m11;
      }
    }
    if (a.stats) {     // sum over the 32 tiles of the block (fixed order), lanes 0 and 32 write their four channels: slab = spatial block
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float u = s1[e], v = s2[e];
#define CRB_ROW_ROR_ADD(x, ctrl) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, false))
        CRB_ROW_ROR_ADD(u, 0x128); CRB_ROW_ROR_ADD(v, 0x128);
        CRB_ROW_ROR_ADD(u, 0x124); CRB_ROW_ROR_ADD(v, 0x124);
        CRB_ROW_ROR_ADD(u, 0x122); CRB_ROW_ROR_ADD(v, 0x122);
        CRB_ROW_ROR_ADD(u, 0x121); CRB_ROW_ROR_ADD(v, 0x121);
#undef CRB_ROW_ROR_ADD
        u += __shfl_xor(u, 16, 64);
        v += __shfl_xor(v, 16, 64);
        s1[e] = u;
        s2[e] = v;
      }
      if (l31 == 0) {
        const int64_t blk = (int64_t)(eu.R0 / TB_ROWS) * a.tw4 + eu.bc;
        float* so = a.stats + (blk * 2) * a.cout + kbase;
        *reinterpret_cast<f32x4*>(so) = s1;
        *reinterpret_cast<f32x4*>(so + a.cout) = s2;
      }
    }
  };
  auto unit_epilogue = [&]() __attribute__((always_inline)) {
    const UnitPos eu = c4::unit_at(e_unit, a);
    const int tr = l31 >> 2, tc = l31 & 3;
    const int Rg = eu.R0 + tr;
    const int tx = eu.bc * TB_COLS + tc;
    int n2 = eu.n0, ty2 = eu.ty0 + tr;
    while (ty2 >= a.th) { ty2 -= a.th; ++n2; }
    const int oy = 2 * ty2, ox = 2 * tx;
    const bool in = Rg < a.RT && tx < a.tw;
    const bool x1 = ox + 1 < a.W, y1 = oy + 1 < a.H;
    float* const yo = a.y + (((int64_t)n2 * a.H + oy) * a.W + ox) * a.cout + eu.cb * WG_K + wave * 32 + 4 * lhi;
    acc_settle();
    epilogue_group(std::integral_constant<int, 0>{}, eu, yo, in, x1, y1);
    epilogue_group(std::integral_constant<int, 1>{}, eu, yo, in, x1, y1);
    epilogue_group(std::integral_constant<int, 2>{}, eu, yo, in, x1, y1);
    epilogue_group(std::integral_constant<int, 3>{}, eu, yo, in, x1, y1);
    acc_zero_range<0, 256>();
    ++e_unit;
  };

  // ---- input transform in pieces placed behind the MFMAs. col<k>: lane-local column k (3 of them) of the 4 x 4 patch,
  //      tp[.][k] = B^T d (the third column times t_sg); xi<r, jj, s>: element jj of the lane's xi pair of row r in five steps
  f32x4 tp[4][3];
  f32x4 sv, sr1, sr2;
  auto t_read = [&](f32x4 (&d)[4][3]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k) d[i][k] = *reinterpret_cast<const f32x4*>(Rb + t_par * c4::RAW_BYTES + raw_rd[k] + i * RAW_PX * 64);
    t_par ^= 1;
  };
  auto t_col = [&](const f32x4 (&d)[4][3], int k) __attribute__((always_inline)) {
    tp[0][k] = d[0][k] - d[2][k];
    tp[1][k] = d[1][k] + d[2][k];
    tp[2][k] = d[2][k] - d[1][k];
    tp[3][k] = d[1][k] - d[3][k];
    if (k == 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) tp[r][2] = tp[r][2] * t_sg;
    }
    asm volatile("" : "+v"(tp[0][k]), "+v"(tp[1][k]), "+v"(tp[2][k]), "+v"(tp[3][k]));
  };
  auto trunc16 = [](float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); };
  auto pack_hi = [](const f32x4& x) {
    u32x2 o;
    o[0] = (__float_as_uint(x[0]) >> 16) | (__float_as_uint(x[1]) & 0xffff0000u);
    o[1] = (__float_as_uint(x[2]) >> 16) | (__float_as_uint(x[3]) & 0xffff0000u);
    return o;
  };
  auto t_xi = [&](unsigned char* V, auto rcst, auto jcst, auto scst) __attribute__((always_inline)) {
    constexpr int r = decltype(rcst)::value, jj = decltype(jcst)::value, s = decltype(scst)::value;
    if constexpr (s == 0) {
      if constexpr (jj == 0) sv = tp[r][0] - tp[r][1];
      else sv = tp[r][1] + tp[r][2];
      asm volatile("" : "+v"(sv));
    } else if constexpr (s == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) sr1[e] = sv[e] - trunc16(sv[e]);
      asm volatile("" : "+v"(sr1));
    } else if constexpr (s == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) sr2[e] = sr1[e] - trunc16(sr1[e]);
      asm volatile("" : "+v"(sr2));
    } else if constexpr (s == 3) {
      *reinterpret_cast<u32x2*>(V + v_wr + (jj * 3 + 0) * V_XP) = pack_hi(sv);
    } else {
      *reinterpret_cast<u32x2*>(V + v_wr + (jj * 3 + 1) * V_XP) = pack_hi(sr1);
      *reinterpret_cast<u32x2*>(V + v_wr + (jj * 3 + 2) * V_XP) = pack_hi(sr2);
    }
  };
  auto t_advance = [&]() {
    if (++tcnt < nch) return;
    tcnt = 0;
    ++t_unit;
    if (t_unit % a.ncb == 0) raw_base(t_unit);
  };

  // ---- phases. f = 4 chunk + i: MFMAs of xi row i on V row i and the fragments requested a phase ago. ONE barrier per TWO xi rows:
  //      rows (0, 1) and (2, 3) are the two V buffers; while the MFMAs run on one, the transform forms the other - row (i + 2) & 3, two
  //      phases ahead, rows 0 and 1 for the next chunk from raw(chunk + 1) (two raw blocks: requested a chunk ahead in phase 2)
  auto phase = [&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    constexpr int rn = (i + 2) & 3;
    if (i == 2) { asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    else if (i == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    unsigned char* const Vn = Vb + rn * V_PHASE;
    const unsigned char* const Vc = Vb + i * V_PHASE + b_rd;
    bf16x8 B[2][3];
    auto op_read = [&](int jx, int slot) __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < 3; ++p) B[slot][p] = *reinterpret_cast<const bf16x8*>(Vc + (jx * 3 + p) * V_XP);
    };
    op_read(0, 0);
    f32x4 d[4][3];
    if (i == 2) { t_read(d); t_advance(); issue_raw(); }
    __builtin_amdgcn_sched_barrier(0);
    // piece k behind MFMA k: the ten steps of the lane's two xi on every other MFMA (phase 3: the three column passes first)
    auto piece = [&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      if constexpr (i == 2 && k < 3) {
        t_col(d, k);
      } else {
        constexpr int k0 = (i == 2) ? k - 3 : k;
        if constexpr (k0 >= 0 && k0 < 20 && (k0 & 1) == 0) {
          t_xi(Vn, std::integral_constant<int, rn>{}, std::integral_constant<int, (k0 >> 1) / 5>{}, std::integral_constant<int, (k0 >> 1) % 5>{});
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto stage = [&](auto jc) __attribute__((always_inline)) {
      constexpr int jx = decltype(jc)::value;
      constexpr int s = jx & 1;
      constexpr int cs = i & 1;
      const bf16x8 &A0 = Ar[cs][jx][0], &A1 = Ar[cs][jx][1], &A2 = Ar[cs][jx][2];
      // loads in flight behind the fragments of (this phase, xi jx), requested two phases ago: the other xi of that phase, the whole
      // phase in between, this phase's requests so far, and the four raw copies of phase 2 while they are younger
      asm volatile("s_waitcnt vmcnt(%0)" : : "n"(i >= 2 ? 25 : 21) : "memory");
      mfma_acc<4 * i + jx>(A0, B[s][2]);
      if (jx < 3) op_read(jx + 1, s ^ 1);
      piece(std::integral_constant<int, 6 * jx + 0>{});
      mfma_acc<4 * i + jx>(A2, B[s][0]);
      piece(std::integral_constant<int, 6 * jx + 1>{});
      mfma_acc<4 * i + jx>(A1, B[s][1]);
      piece(std::integral_constant<int, 6 * jx + 2>{});
      mfma_acc<4 * i + jx>(A0, B[s][1]);
      piece(std::integral_constant<int, 6 * jx + 3>{});
      mfma_acc<4 * i + jx>(A1, B[s][0]);
      piece(std::integral_constant<int, 6 * jx + 4>{});
      mfma_acc<4 * i + jx>(A0, B[s][0]);
      a_load(std::integral_constant<int, cs>{}, jc);
      if (jx == 3) u_advance();
      piece(std::integral_constant<int, 6 * jx + 5>{});
    };
    stage(std::integral_constant<int, 0>{});
    stage(std::integral_constant<int, 1>{});
    stage(std::integral_constant<int, 2>{});
    stage(std::integral_constant<int, 3>{});
    if (i == 2) r_advance();
  };

  // ---- prologue: raw(0) and the fragments of phase 0 in one round trip, raw(0) -> tp -> V(0)
  issue_raw(); r_advance();
  issue_raw(); r_advance();                                       // raw(0), raw(1)
  {
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    a_load(I0{}, I0{}); a_load(I0{}, I1{}); a_load(I0{}, I2{}); a_load(I0{}, I3{});
    u_advance();
    a_load(I1{}, I0{}); a_load(I1{}, I1{}); a_load(I1{}, I2{}); a_load(I1{}, I3{});
    u_advance();
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
    f32x4 d[4][3];
    t_read(d); t_advance();
#pragma unroll
    for (int k = 0; k < 3; ++k) t_col(d, k);
  }
#define CRB_T_XI_ALL(JJ)                                                                                                      \
  t_xi(Vb, std::integral_constant<int, 0>{}, std::integral_constant<int, JJ>{}, std::integral_constant<int, 0>{});            \
  t_xi(Vb, std::integral_constant<int, 0>{}, std::integral_constant<int, JJ>{}, std::integral_constant<int, 1>{});            \
  t_xi(Vb, std::integral_constant<int, 0>{}, std::integral_constant<int, JJ>{}, std::integral_constant<int, 2>{});            \
  t_xi(Vb, std::integral_constant<int, 0>{}, std::integral_constant<int, JJ>{}, std::integral_constant<int, 3>{});            \
  t_xi(Vb, std::integral_constant<int, 0>{}, std::integral_constant<int, JJ>{}, std::integral_constant<int, 4>{})
  CRB_T_XI_ALL(0); CRB_T_XI_ALL(1);
#undef CRB_T_XI_ALL
#define CRB_T_XI_ALL1(JJ)                                                                                                                 \
  t_xi(Vb + V_PHASE, std::integral_constant<int, 1>{}, std::integral_constant<int, JJ>{}, std::integral_constant<int, 0>{});            \
  t_xi(Vb + V_PHASE, std::integral_constant<int, 1>{}, std::integral_constant<int, JJ>{}, std::integral_constant<int, 1>{});            \
  t_xi(Vb + V_PHASE, std::integral_constant<int, 1>{}, std::integral_constant<int, JJ>{}, std::integral_constant<int, 2>{});            \
  t_xi(Vb + V_PHASE, std::integral_constant<int, 1>{}, std::integral_constant<int, JJ>{}, std::integral_constant<int, 3>{});            \
  t_xi(Vb + V_PHASE, std::integral_constant<int, 1>{}, std::integral_constant<int, JJ>{}, std::integral_constant<int, 4>{})
  CRB_T_XI_ALL1(0); CRB_T_XI_ALL1(1);
#undef CRB_T_XI_ALL1

  for (int cg = 0; cg < total_chunks; ++cg) {
    phase(std::integral_constant<int, 0>{});
    phase(std::integral_constant<int, 1>{});
    phase(std::integral_constant<int, 2>{});
    phase(std::integral_constant<int, 3>{});
    if (++ec == nch) { ec = 0; unit_epilogue(); }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (fragments requested past the end of the range)
}

#ifdef CRB_MEASURE   // the second form: A/B in the measurement library (it is not faster: see its header)
// ================================================================================================================================
// Second form (measurement library, crb_winograd4_set_variant(2)): the same pipeline with TWO waves per SIMD. Measured on the first form above (one 512-register wave
// per SIMD, profiles/r06_time_winograd4_v1_skip_work_modes.txt): the skip-work builds add up - MFMAs 140 us, LDS-DMA 130 us,
// operand reads 90 us, transform 75 us, V stores 50 us, output stores 45 us of 635 us: with one in-order wave per SIMD nothing
// runs under anything else. Here a workgroup is 512 threads = 8 waves of 128 accumulators + 128 registers: wave = 32 tiles x 32
// output channels x HALF of the xi (columns 2 xh, 2 xh + 1 of every xi row), so each SIMD has a second wave to issue from while
// one waits for LDS, for the matrix pipe or at a counter. Price: the output transform needs both halves. Y = A^T M A splits as
// Y = Q(xh = 0) + Q(xh = 1) with Q = A^T (M_half A_half) formed by each wave on its own accumulators; wave xh keeps output row xh
// of its Q and hands the other row to its partner (wave ^ 4) through LDS (the raw block and the V image that are idle between
// two units: 2 rounds x 32 KB, three extra barriers per unit).
constexpr int NT2 = 512;
// The 128 accumulators are a16 .. a143 (xi (row i, column 2 xh + jj) -> a[16 + 16 (2 i + jj) ..]); a0 .. a15 are left to the compiler, which
// parks values there (mostly the address arithmetic of raw_sources, once per spatial block) when the 112 VGPRs it is given run out
// (amdgpu_num_vgpr: 112 + 144 = 256 = two waves per SIMD). tools/check_wino4_isa.py fails the build if it ever writes a VGPR into a16 or above.
constexpr int ACC0 = 16;
template <int XI>
__device__ __forceinline__ void mfma_acc8(const bf16x8& A, const bf16x8& B) {
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%2:%3], %0, %1, a[%2:%3]" : : "v"(A), "v"(B), "n"(ACC0 + XI * 16), "n"(ACC0 + XI * 16 + 15));
}

template <int MODE>
__global__ __launch_bounds__(NT2, 2) __attribute__((amdgpu_num_vgpr(112))) void winograd4b_kernel(Wino4Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const Vb = lds + LDS_V;
  unsigned char* const Ub = lds + LDS_U;
  unsigned char* const Rb = lds + LDS_RAW;
  // measurement builds: MODE 1 .. 9 as listed above; MODE >= 64: a bit mask of what is LEFT OUT (1 MFMAs, 2 transform, 4 copies, 8 operand
  // reads, 16 output stores) - "what does this ingredient cost beside the MFMAs alone"
  constexpr int FL = MODE >= 64 ? MODE - 64 : 0;
  constexpr bool NO_MFMA = MODE == 1 || (FL & 1), NO_T = MODE == 2 || (FL & 2), NO_DMA = MODE == 3 || (FL & 4), NO_OPR = MODE == 4 || (FL & 8),
                 NO_OUT = MODE == 8 || (FL & 16);
  const int T = threadIdx.x, lane = T & 63, wave = __builtin_amdgcn_readfirstlane(T >> 6);
  asm volatile("" ::: "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143");

  // ---- the unit range of this workgroup
  const int nunits = a.nblocks * a.ncb;
  int G = gridDim.x;
  if (a.seq) {
    if (T == 0) {
      unsigned* L = g_cu_latch4 + (a.seq & 63u);
      unsigned v = __hip_atomic_load(L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), mine;
      for (;;) {
        if ((v >> 8) == a.seq) { mine = v & 255u; break; }
        const int b = __hip_atomic_load(&g_cu_busy4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned busy = (unsigned)min(max(b, 0), 255);
        const unsigned seen = atomicCAS(L, v, (a.seq << 8) | busy);
        if (seen == v) { mine = busy; break; }
        v = seen;
      }
      *reinterpret_cast<unsigned*>(Rb) = mine;
    }
    __syncthreads();
    const int busy = (int)*reinterpret_cast<const unsigned*>(Rb);
    __syncthreads();
    G = max(1, (int)gridDim.x - __builtin_amdgcn_readfirstlane(busy));
    if ((int)blockIdx.x >= G) return;
  }
  const int u_first = (int)((int64_t)blockIdx.x * nunits / G);
  const int u_end = (int)((int64_t)(blockIdx.x + 1) * nunits / G);
  if (u_first >= u_end) return;
  const int nch = a.cin / CC;
  const int total_chunks = (u_end - u_first) * nch;

  // ---- transform role: thread = (tile, channel pair)
  const int t_tile = T >> 3, t_cp = T & 7, t_tr = t_tile >> 2, t_tc = t_tile & 3;
  // V store (4 bytes = the pair's two bf16): [xi][piece][k group = pair >> 2][tile][pair & 3] (+ (jx * 3 + p) * V_XP)
  const int v_wr = (t_cp >> 2) * V_REGION + t_tile * 16 + (t_cp & 3) * 4;
  int t_unit = u_first, tcnt = 0;                   // unit and chunk of it that the NEXT t_read reads
  auto raw_base = [&](int u) {
    const UnitPos p = unit_at(u, a);
    const int rows_a = min(TB_ROWS, a.th - p.ty0);           // tile rows of the block that belong to its first image
    const int lr = 2 * t_tr + (t_tr >= rows_a ? 2 : 0);
    return 64 * (lr * RAW_PX + t_tc) + 8 * t_cp;
  };
  int raw_rd = raw_base(t_unit);

  // ---- DMA role. raw: slots s = T + 512 j (16 bytes each): s -> (pixel position P = s >> 2, channel quad s & 3), P = local
  //      row * 10 + column order (x >> 1) + 5 (x & 1). U: 3 x 16 bytes per thread and phase.
  int r_unit = u_first, rc = 0;
  const float* rsrc[3];
  unsigned rstep = 0;
  auto raw_sources = [&](int u) {
    const UnitPos p = unit_at(u, a);
    const int rows_a = min(TB_ROWS, a.th - p.ty0);
    const int limit_a = 2 * rows_a + 2;
    rstep = 0;
    int Tq = T;
    asm volatile("" : "+v"(Tq));        // (opaque: the slot arithmetic below is redone per spatial block instead of living in registers
                                        // across the chunk loop, where the wave has none to spare)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int s = Tq + NT2 * j;
      const int P = s >> 2, q = s & 3;
      const int lr = P / RAW_PX, rem = P - lr * RAW_PX;
      const int xx = rem < 5 ? 2 * rem : 2 * (rem - 5) + 1;
      const bool in_a = lr < limit_a;
      const int n = in_a ? p.n0 : p.n0 + 1;
      const int py = in_a ? 2 * p.ty0 - 1 + lr : lr - limit_a - 1;
      const int px = 8 * p.bc - 1 + xx;
      const bool ok = lr < RAW_ROWS && (in_a || (rows_a < TB_ROWS && (lr - limit_a) < 2 * (TB_ROWS - rows_a) + 2)) && n < a.N &&
                      py >= 0 && py < a.H && px >= 0 && px < a.W;
      const int64_t off = (((int64_t)n * a.H + py) * a.W + px) * a.cin + q * 4;
      rsrc[j] = ok ? a.x + off : g_wino4_zero_page;
      rstep |= ok ? (1u << j) : 0u;
    }
  };
  raw_sources(r_unit);
  auto issue_raw = [&]() {
#pragma unroll
    for (int j = 0; j < 3; ++j) glds16(rsrc[j], Rb + (NT2 * j + wave * 64) * 16);
  };
  auto r_advance = [&]() {
    if (++rc < nch) {
#pragma unroll
      for (int j = 0; j < 3; ++j) rsrc[j] += (rstep >> j & 1u) ? CC : 0;
      return;
    }
    rc = 0;
    ++r_unit;
    if (r_unit % a.ncb == 0) raw_sources(r_unit);            // next spatial block (channel block 0 again)
    else {
#pragma unroll
      for (int j = 0; j < 3; ++j) rsrc[j] -= (rstep >> j & 1u) ? (nch - 1) * CC : 0;
    }
  };
  int u_cb = u_first % a.ncb, upc = 0;
  const unsigned char* usrc = a.U + (int64_t)u_cb * nch * 4 * U_PHASE + T * 16;
  auto issue_u = [&](int ub) {
#pragma unroll
    for (int k = 0; k < 3; ++k) glds16(usrc + k * (NT2 * 16), Ub + ub + (k * NT2 + wave * 64) * 16);
  };
  auto u_advance = [&]() {
    if (++upc < nch * 4) { usrc += U_PHASE; return; }
    upc = 0;
    if (++u_cb == a.ncb) u_cb = 0;
    usrc = a.U + (int64_t)u_cb * nch * 4 * U_PHASE + T * 16;
  };

  // ---- MFMA role: wave = tile half (wave & 1) x channel half ((wave >> 1) & 1) x xi-column half (wave >> 2)
  const int w_th = wave & 1, w_kh = (wave >> 1) & 1, w_xh = wave >> 2;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int a_rd = w_xh * 6 * U_XP + lhi * (64 * 16) + (w_kh * 32 + l31) * 16;       // U image: A operand, rows = output channels
  const int b_rd = w_xh * 6 * V_XP + lhi * V_REGION + (w_th * 32 + l31) * 16;        // V image: B operand, columns = tiles
  acc_zero_range<ACC0, 128>();

  // ---- output transform of a finished unit (see the head of this kernel). Accumulator of xi (row i, column 2 xh + jj): a[(2 i + jj)
  //      * 16 ..]; lane = tile l31 of the wave's tile half, channels cb * 64 + 32 kh + 8 j + 4 lhi + (0..3) for register group j
  int e_unit = u_first, ec = 0;
  auto unit_epilogue = [&]() __attribute__((always_inline)) {
    const UnitPos eu = unit_at(e_unit, a);
    const int tile = w_th * 32 + l31;
    const int tr = tile >> 2, tc = tile & 3;
    const int Rg = eu.R0 + tr;
    const int tx = eu.bc * TB_COLS + tc;
    int n2 = eu.n0, ty2 = eu.ty0 + tr;
    while (ty2 >= a.th) { ty2 -= a.th; ++n2; }
    const int oy = 2 * ty2 + w_xh, ox = 2 * tx;                   // this wave finalises output row xh of the 2 x 2
    const bool in = Rg < a.RT && tx < a.tw && oy < a.H;
    const bool x1 = ox + 1 < a.W;
    const int kb = eu.cb * WG_K + w_kh * 32 + 4 * lhi;
    float* const yo = a.y + (((int64_t)n2 * a.H + oy) * a.W + ox) * a.cout + kb;
    unsigned char* const X0 = Rb;                                 // exchange areas: [column 2][wave 8][lane 64] 16 bytes = 16 KB each
    unsigned char* const X1 = Ub + 2 * U_PHASE;                   // (the third U image of the first form: this pipeline runs on two)
    const int x_wr = wave * 1024 + lane * 16, x_rd = (wave ^ 4) * 1024 + lane * 16;
    acc_settle();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                 // every wave is done with the raw block
    auto group_q = [&](auto jc, f32x4 (&keep)[2], unsigned char* X) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      // rows first, streaming over the accumulators: S0[x] = M[0][x] + M[1][x] + M[2][x], S1[x] = M[1][x] - M[2][x] - M[3][x] for the
      // wave's two columns x (A^T = [1 1 1 0; 0 1 -1 -1] from the left), then the columns: xh = 0 holds columns 0, 1 of M
      // (Q[.][0] = S[0] + S[1], Q[.][1] = S[1]), xh = 1 columns 2, 3 (Q[.][0] = S[0], Q[.][1] = -S[0] - S[1])
#define CRB_ACC4(I, JJ) ((f32x4){acc_read<ACC0 + (2 * I + JJ) * 16 + 4 * j>(), acc_read<ACC0 + (2 * I + JJ) * 16 + 4 * j + 1>(), \
                                 acc_read<ACC0 + (2 * I + JJ) * 16 + 4 * j + 2>(), acc_read<ACC0 + (2 * I + JJ) * 16 + 4 * j + 3>()})
      f32x4 S0[2], S1[2];
      {
        const f32x4 m00 = CRB_ACC4(0, 0), m01 = CRB_ACC4(0, 1), m10 = CRB_ACC4(1, 0), m11 = CRB_ACC4(1, 1);
        S0[0] = m00 + m10; S0[1] = m01 + m11;
        S1[0] = m10; S1[1] = m11;
      }
      {
        const f32x4 m20 = CRB_ACC4(2, 0), m21 = CRB_ACC4(2, 1), m30 = CRB_ACC4(3, 0), m31 = CRB_ACC4(3, 1);
        S0[0] = S0[0] + m20; S0[1] = S0[1] + m21;
        S1[0] = S1[0] - m20 - m30; S1[1] = S1[1] - m21 - m31;
      }
#undef CRB_ACC4
      const f32x4 q00 = w_xh ? S0[0] : S0[0] + S0[1], q01 = w_xh ? -S0[0] - S0[1] : S0[1];
      const f32x4 q10 = w_xh ? S1[0] : S1[0] + S1[1], q11 = w_xh ? -S1[0] - S1[1] : S1[1];
      keep[0] = w_xh ? q10 : q00;
      keep[1] = w_xh ? q11 : q01;
      *reinterpret_cast<f32x4*>(X + x_wr) = w_xh ? q00 : q10;              // the partner's row
      *reinterpret_cast<f32x4*>(X + 8 * 1024 + x_wr) = w_xh ? q01 : q11;
    };
    auto group_out = [&](auto jc, const f32x4 (&keep)[2], const unsigned char* X) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      const f32x4 o0 = *reinterpret_cast<const f32x4*>(X + x_rd), o1 = *reinterpret_cast<const f32x4*>(X + 8 * 1024 + x_rd);
      f32x4 bias = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (a.bias) {
        const float* bp = a.bias + eu.cb * WG_K + __builtin_amdgcn_readfirstlane(w_kh) * 32 + 8 * j;
        const f32x4 b0 = sload4(bp), b1 = sload4(bp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) bias[e] = lhi ? b1[e] : b0[e];
      }
      // (xi-column half 0 first in both waves: the two rows of an output are formed by the same expression)
      f32x4 y0 = (w_xh ? o0 + keep[0] : keep[0] + o0) + bias, y1 = (w_xh ? o1 + keep[1] : keep[1] + o1) + bias;
      if (a.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { y0[e] = fmaxf(y0[e], 0.f); y1[e] = fmaxf(y1[e], 0.f); }
      }
      f32x4 s1 = (f32x4){0.f, 0.f, 0.f, 0.f}, s2 = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (in && !NO_OUT) {
        float* o = yo + 8 * j;
        *reinterpret_cast<f32x4*>(o) = y0;
        if (x1) *reinterpret_cast<f32x4*>(o + a.cout) = y1;
      }
      if (a.stats) {
        if (in) {
          const float m1 = x1 ? 1.f : 0.f;
          s1 = y0 + y1 * m1;
          s2 = y0 * y0 + (y1 * y1) * m1;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float u = s1[e], v = s2[e];
#define CRB_ROW_ROR_ADD(x, ctrl) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, false))
          CRB_ROW_ROR_ADD(u, 0x128); CRB_ROW_ROR_ADD(v, 0x128);       // row_ror:8
          CRB_ROW_ROR_ADD(u, 0x124); CRB_ROW_ROR_ADD(v, 0x124);
          CRB_ROW_ROR_ADD(u, 0x122); CRB_ROW_ROR_ADD(v, 0x122);
          CRB_ROW_ROR_ADD(u, 0x121); CRB_ROW_ROR_ADD(v, 0x121);
#undef CRB_ROW_ROR_ADD
          u += __shfl_xor(u, 16, 64);
          v += __shfl_xor(v, 16, 64);
          s1[e] = u;
          s2[e] = v;
        }
        if (l31 == 0) {                      // slab = (spatial block, tile half, output row)
          const int64_t blk = (int64_t)(eu.R0 / TB_ROWS) * a.tw4 + eu.bc;
          float* so = a.stats + (((blk * 2 + w_th) * 2 + w_xh) * 2) * a.cout + kb + 8 * j;
          *reinterpret_cast<f32x4*>(so) = s1;
          *reinterpret_cast<f32x4*>(so + a.cout) = s2;
        }
      }
    };
    f32x4 k0[2], k1[2];
    group_q(std::integral_constant<int, 0>{}, k0, X0);
    group_q(std::integral_constant<int, 1>{}, k1, X1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    group_out(std::integral_constant<int, 0>{}, k0, X0);
    group_out(std::integral_constant<int, 1>{}, k1, X1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                 // the partners have read round 0
    group_q(std::integral_constant<int, 2>{}, k0, X0);
    group_q(std::integral_constant<int, 3>{}, k1, X1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    group_out(std::integral_constant<int, 2>{}, k0, X0);
    group_out(std::integral_constant<int, 3>{}, k1, X1);
    acc_zero_range<ACC0, 128>();
    ++e_unit;
    // (the next phase's barrier orders these exchange reads in front of the raw copy and the V stores that reuse the areas)
  };

  unsigned long long st_wait = 0, st_bar = 0, st_head = 0, st_body = 0, st_epi = 0, st_start = 0;      // MODE 9: s_memtime sums
  // ---- input transform pieces (see the first form): col<b> (its four raw values are read one piece earlier), and per xi two steps:
  //      [value, remainders] and [packs + 3 stores]
  f32x2 tp[4][4];
  f32x2 sv, sr1, sr2;
  auto t_col = [&](int b) __attribute__((always_inline)) {       // column b of the patch: 4 raw values -> tp[.][b] (the SIMD's other wave covers the read)
    if (NO_T) return;
    const unsigned char* p = Rb + raw_rd + ((b >> 1) + 5 * (b & 1)) * 64;      // pixel column b: position t_tc + (b >> 1) + 5 (b & 1)
    f32x2 d[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = *reinterpret_cast<const f32x2*>(p + i * RAW_PX * 64);
    tp[0][b] = d[0] - d[2];
    tp[1][b] = d[1] + d[2];
    tp[2][b] = d[2] - d[1];
    tp[3][b] = d[1] - d[3];
    asm volatile("" : "+v"(tp[0][b]), "+v"(tp[1][b]), "+v"(tp[2][b]), "+v"(tp[3][b]));
  };
  auto trunc16 = [](float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); };
  auto pack_hi = [](const f32x2& x) { return (__float_as_uint(x[0]) >> 16) | (__float_as_uint(x[1]) & 0xffff0000u); };
  auto t_xi = [&](unsigned char* V, auto rcst, auto jcst, auto scst) __attribute__((always_inline)) {
    constexpr int r = decltype(rcst)::value, jx = decltype(jcst)::value, s = decltype(scst)::value;
    if (NO_T) return;
    if constexpr (s == 0) {
      if constexpr (jx == 0) sv = tp[r][0] - tp[r][2];
      else if constexpr (jx == 1) sv = tp[r][1] + tp[r][2];
      else if constexpr (jx == 2) sv = tp[r][2] - tp[r][1];
      else sv = tp[r][1] - tp[r][3];
#pragma unroll
      for (int e = 0; e < 2; ++e) sr1[e] = sv[e] - trunc16(sv[e]);
#pragma unroll
      for (int e = 0; e < 2; ++e) sr2[e] = sr1[e] - trunc16(sr1[e]);
      asm volatile("" : "+v"(sv), "+v"(sr1), "+v"(sr2));
    } else {
      if (MODE == 7) { asm volatile("" :: "v"(pack_hi(sv)), "v"(pack_hi(sr1)), "v"(pack_hi(sr2))); return; }
      *reinterpret_cast<unsigned*>(V + v_wr + (jx * 3 + 0) * V_XP) = pack_hi(sv);
      *reinterpret_cast<unsigned*>(V + v_wr + (jx * 3 + 1) * V_XP) = pack_hi(sr1);
      *reinterpret_cast<unsigned*>(V + v_wr + (jx * 3 + 2) * V_XP) = pack_hi(sr2);
    }
  };
  auto t_advance = [&]() {
    if (++tcnt < nch) return;
    tcnt = 0;
    ++t_unit;
    if (t_unit % a.ncb == 0) raw_rd = raw_base(t_unit);
  };

  // ---- phases, software-pipelined so that NOTHING but the counter wait and the barrier stands between two phases' MFMAs (the second
  //      form's first build read its operands and issued its copies between the barrier and the first MFMA, in all eight waves at
  //      once: matrix pipe 23 % busy). Phase f = 4 chunk + i runs the 12 MFMAs of xi row i on operands that are ALREADY in registers
  //      and, one piece behind every MFMA: requests U(f + 2) -> U[f & 1] (and raw(chunk + 1) in phase 0), forms row (i + 2) & 3 of
  //      V(f + 2) -> V[f & 1] (phase 2 first reads the new raw block and does the column pass), and reads the operands of phase f + 1
  //      from V[(f + 1) & 1], U[(f + 1) & 1] into the registers the MFMAs have just released. The barrier at the head of phase f
  //      certifies V(f + 1), U(f + 1) and that every wave is done reading V(f), U(f).
  bf16x8 A[2][3], B[2][3];                           // operands of the running phase: [xi jj][piece]
  auto op_read = [&](int f1, int p, bool is_a) __attribute__((always_inline)) {      // piece p of both xi of the phase with parity f1
    if (NO_OPR) return;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      if (is_a) A[jj][p] = *reinterpret_cast<const bf16x8*>(Ub + f1 * U_PHASE + a_rd + (jj * 3 + p) * U_XP);
      else B[jj][p] = *reinterpret_cast<const bf16x8*>(Vb + f1 * V_PHASE + b_rd + (jj * 3 + p) * V_XP);
    }
  };
  auto phase = [&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    constexpr int rn = (i + 2) & 3;
    constexpr int f0 = i & 1, f1 = f0 ^ 1;             // parity of f (buffers being refilled) and of f + 1 (buffers being read)
    unsigned long long tm0 = 0, tm1 = 0, tm2 = 0;
    if (MODE == 9) tm0 = __builtin_amdgcn_s_memtime();
    // copies that may stay in flight: raw(chunk + 1), requested in phase 0 behind U(f + 2) and read in phase 2
    if (i == 1) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (MODE == 9) tm1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();
    if (MODE == 9) { tm2 = __builtin_amdgcn_s_memtime(); st_wait += tm1 - tm0; st_bar += tm2 - tm1; }
    unsigned char* const Vn = Vb + f0 * V_PHASE;
    // transform piece t of the phase (phase 2: the column passes first, columns 0, 2 first: xi 0 needs them)
    auto tpiece = [&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
      constexpr int t0 = (i == 2) ? t - 4 : t;
      if constexpr (i == 2 && t < 4) {
        constexpr int order[4] = {0, 2, 1, 3};
        t_col(order[t]);
        if constexpr (t == 3) t_advance();
      } else if constexpr (t0 >= 0 && t0 < 8) {
        t_xi(Vn, std::integral_constant<int, rn>{}, std::integral_constant<int, t0 / 2>{}, std::integral_constant<int, t0 % 2>{});
      }
    };
    // slot k: the work placed behind MFMA k
    auto slot = [&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      if constexpr (k == 0) {
        if (!NO_DMA) {
          if (MODE != 5) { issue_u(f0 * U_PHASE); u_advance(); }
          if (i == 0 && MODE != 6) { issue_raw(); r_advance(); }
        }
      }
      if constexpr (k == 1) op_read(f1, 2, false);       // B piece 2 was used by MFMAs 0, 1
      if constexpr (k == 5) op_read(f1, 1, false);       // B piece 1: MFMAs 2 .. 5
      if constexpr (k == 7) op_read(f1, 0, true);        // A piece 0: MFMAs 0 .. 3, 6, 7
      if constexpr (k == 9) op_read(f1, 1, true);        // A piece 1: MFMAs 4, 5, 8, 9
      if constexpr (k == 11) { op_read(f1, 2, true); op_read(f1, 0, false); }
      if constexpr (i == 2) {                            // 12 transform pieces over 12 slots
        tpiece(std::integral_constant<int, k>{});
      } else if constexpr (k >= 1 && k <= 8) {
        tpiece(std::integral_constant<int, k - 1>{});
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 9) { tm0 = __builtin_amdgcn_s_memtime(); st_head += tm0 - tm2; }
#define CRB_MM(K, JJ, PA, PB)                                      \
    if (!NO_MFMA) mfma_acc8<2 * i + JJ>(A[JJ][PA], B[JJ][PB]);    \
    slot(std::integral_constant<int, K>{})
    CRB_MM(0, 0, 0, 2); CRB_MM(1, 1, 0, 2);
    CRB_MM(2, 0, 0, 1); CRB_MM(3, 1, 0, 1);
    CRB_MM(4, 0, 1, 1); CRB_MM(5, 1, 1, 1);
    CRB_MM(6, 0, 0, 0); CRB_MM(7, 1, 0, 0);
    CRB_MM(8, 0, 1, 0); CRB_MM(9, 1, 1, 0);
    CRB_MM(10, 0, 2, 0); CRB_MM(11, 1, 2, 0);
#undef CRB_MM
    if (MODE == 9) st_body += __builtin_amdgcn_s_memtime() - tm0;
  };

  // ---- prologue: raw(0), U(0), U(1) in one round trip; raw(0) -> tp -> rows 0, 1 = V(0), V(1); operands of phase 0
  if (MODE == 9) st_start = __builtin_amdgcn_s_memtime();
  issue_raw(); r_advance();
  issue_u(0); u_advance();
  issue_u(U_PHASE); u_advance();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int b = 0; b < 4; ++b) t_col(b);
  t_advance();
#define CRB_T_XI_ALL(V, R, JX)                                                                                                \
  t_xi(V, std::integral_constant<int, R>{}, std::integral_constant<int, JX>{}, std::integral_constant<int, 0>{});             \
  t_xi(V, std::integral_constant<int, R>{}, std::integral_constant<int, JX>{}, std::integral_constant<int, 1>{})
  CRB_T_XI_ALL(Vb, 0, 0); CRB_T_XI_ALL(Vb, 0, 1); CRB_T_XI_ALL(Vb, 0, 2); CRB_T_XI_ALL(Vb, 0, 3);
  CRB_T_XI_ALL(Vb + V_PHASE, 1, 0); CRB_T_XI_ALL(Vb + V_PHASE, 1, 1); CRB_T_XI_ALL(Vb + V_PHASE, 1, 2); CRB_T_XI_ALL(Vb + V_PHASE, 1, 3);
#undef CRB_T_XI_ALL
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int p = 0; p < 3; ++p) { op_read(0, p, true); op_read(0, p, false); }
  if (NO_OPR) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int p = 0; p < 3; ++p) asm volatile("" : "=v"(A[jj][p]), "=v"(B[jj][p]));
  }

  for (int cg = 0; cg < total_chunks; ++cg) {
    phase(std::integral_constant<int, 0>{});
    phase(std::integral_constant<int, 1>{});
    phase(std::integral_constant<int, 2>{});
    phase(std::integral_constant<int, 3>{});
    if (++ec == nch) {
      ec = 0;
      unsigned long long te = 0;
      if (MODE == 9) te = __builtin_amdgcn_s_memtime();
      unit_epilogue();
      // the operands of the next phase, read during phase 3, were not kept across the output transform (48 registers): again
#pragma unroll
      for (int p = 0; p < 3; ++p) { op_read(0, p, true); op_read(0, p, false); }
      if (MODE == 9) st_epi += __builtin_amdgcn_s_memtime() - te;
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (copies requested past the end of the range)
  if (MODE == 9 && g_wino4_dbg && lane == 0) {
    unsigned long long* o = g_wino4_dbg + ((int64_t)blockIdx.x * 8 + wave) * 8;
    o[0] = st_wait; o[1] = st_bar; o[2] = st_head; o[3] = st_body; o[4] = st_epi; o[5] = __builtin_amdgcn_s_memtime() - st_start;
    o[6] = (unsigned long long)total_chunks; o[7] = (unsigned long long)(u_end - u_first);
  }
}

#endif  // CRB_MEASURE (second form)

}  // namespace

CRB_KNOB g_wino4_mode [[maybe_unused]] = 0;      // measurement builds: see MODE
CRB_KNOB g_wino4_variant [[maybe_unused]] = 1;   // 1 = one 512-register wave per SIMD, U in registers (the product kernel), 2 = the second form, 3 = the first form with U through LDS-DMA (measurement library)
#ifdef CRB_MEASURE
extern "C" int crb_winograd4_set_debug(void* dev_buf) {
  unsigned long long* p = (unsigned long long*)dev_buf;
  CRB_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wino4_dbg), &p, sizeof(p)));
  return CRB_OK;
}
extern "C" int crb_winograd4_set_variant(int v) { g_wino4_variant = (v == 2 || v == 3) ? v : 1; return CRB_OK; }
extern "C" int crb_winograd4_set_mode(int mode) { g_wino4_mode = ((mode >= 1 && mode <= 9) || (mode >= 64 && mode < 96)) ? mode : 0; return CRB_OK; }
#endif

// ceil(H / 2) >= 16: a block of 16 tile rows touches at most two images (the raw block keeps ONE gap)
extern "C" int crb_winograd4_supported(int cin, int cout, int H, int W) {
  return (cin > 0 && cout > 0 && cin % CC == 0 && cout % WG_K == 0 && H >= 31 && W >= 1) ? 1 : 0;
}

// the 32-tile x 128-channel form: ceil(H / 2) >= 8 keeps a block of 8 tile rows inside two images
extern "C" int crb_winograd4c_supported(int cin, int cout, int H, int W) {
  return (cin > 0 && cout > 0 && cin % CC == 0 && cout % c4::WG_K == 0 && H >= 31 && W >= 1) ? 1 : 0;
}

extern "C" int64_t crb_winograd4_weights_bytes(int cin, int cout) { return (int64_t)16 * cin * cout * 6; }

// w = nn.Conv2d weight (Cout,Cin,3,3) f32 with element strides (so, si, sky, skx); mode 0: image of the forward convolution
// (Cin -> Cout), mode 1: image of the input-gradient convolution (Cout -> Cin, flipped / transposed weights)
// (mode + 2: the same image in the layout of the 32-tile x 128-channel form, crb_winograd4c_supported shapes)
extern "C" int crb_winograd4_weights_conv(const float* w, int64_t so, int64_t si, int64_t sky, int64_t skx, void* U, int conv_cin,
                                          int conv_cout, int mode, void* stream) {
  if (mode < 0 || mode > 3) return CRB_ERR_ARG;
  const int kin = (mode & 1) ? conv_cout : conv_cin, kout = (mode & 1) ? conv_cin : conv_cout;
  if (!((mode & 2) ? crb_winograd4c_supported(kin, kout, 31, 1) : crb_winograd4_supported(kin, kout, 31, 1))) return CRB_ERR_UNSUPPORTED;
  const int64_t per = (int64_t)(kin / 8) * kout;
  hipLaunchKernelGGL(winograd4_weights_conv_kernel, dim3(crb_cdiv(per, 256)), dim3(256), 0, (hipStream_t)stream, w, so, si, sky, skx,
                     (unsigned char*)U, kin, kout, mode);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_winograd4_weights_conv_multi(int n, const float* const* w, const int64_t* strides, void* const* U,
                                                const int32_t* conv_cin, const int32_t* conv_cout, const int32_t* mode, void* stream) {
  if (n < 0 || n > WJ_MAX || (n > 0 && (!w || !strides || !U || !conv_cin || !conv_cout || !mode))) return CRB_ERR_ARG;
  if (n == 0) return CRB_OK;
  Wino4WJobs jobs;
  jobs.n = n;
  int64_t blocks = 0;
  for (int j = 0; j < n; ++j) {
    if (mode[j] < 0 || mode[j] > 3) return CRB_ERR_ARG;
    const int kin = (mode[j] & 1) ? conv_cout[j] : conv_cin[j], kout = (mode[j] & 1) ? conv_cin[j] : conv_cout[j];
    if (!w[j] || !U[j]) return CRB_ERR_ARG;
    if (!((mode[j] & 2) ? crb_winograd4c_supported(kin, kout, 31, 1) : crb_winograd4_supported(kin, kout, 31, 1))) return CRB_ERR_UNSUPPORTED;
    jobs.job[j] = Wino4WJob{w[j], (unsigned char*)U[j], strides[4 * j], strides[4 * j + 1], strides[4 * j + 2], strides[4 * j + 3], kin,
                            kout, mode[j], (int)blocks};
    blocks += crb_cdiv((int64_t)(kin / 8) * kout, 256);
    if (blocks >= (1LL << 30)) return CRB_ERR_ARG;
  }
  hipLaunchKernelGGL(winograd4_weights_conv_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, jobs);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

namespace {
constexpr int MAX_DEV = 64;
std::atomic<int> g_dev_cus4[MAX_DEV];
std::atomic<unsigned> g_dev_attr4[MAX_DEV];
std::atomic<unsigned> g_dev_seq4[MAX_DEV];                     // launch sequence PER DEVICE: the latch ring (64 slots) lives in that device's memory,
                                                             // two launches share a slot only if 64 launches to the SAME device lie between them

int device_cus4(int* dev_out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return -1;
  *dev_out = dev;
  int n = g_dev_cus4[dev].load(std::memory_order_relaxed);
  if (n > 0) return n;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
  n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  g_dev_cus4[dev].store(n, std::memory_order_relaxed);
  return n;
}
}  // namespace

// (called by crb_cu_reservation in winograd_conv2.hip: every persistent Winograd kernel sees the announcement)
int crbhip_wino4_cu_busy_set(int cus, hipStream_t stream) {
  hipLaunchKernelGGL(cu_busy4_set_kernel, dim3(1), dim3(1), 0, stream, cus);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

static int winograd4c_launch(const float* x, const void* U, float* y, int N, int H, int W, int cin, int cout, const float* bias, int relu,
                             void* stream, float* stats) {
  if (N <= 0 || H <= 0 || W <= 0) return CRB_ERR_ARG;
  if (!crb_winograd4c_supported(cin, cout, H, W)) return CRB_ERR_UNSUPPORTED;
  Wino4Args a;
  a.x = x; a.U = (const unsigned char*)U; a.y = y; a.bias = bias; a.stats = stats;
  a.N = N; a.H = H; a.W = W; a.cin = cin; a.cout = cout; a.relu = relu;
  a.th = (H + 1) / 2; a.tw = (W + 1) / 2;
  const int64_t rt = (int64_t)N * a.th;
  if (rt >= (1LL << 30) || (int64_t)N * H * W * (cin > cout ? cin : cout) >= (1LL << 40)) return CRB_ERR_ARG;
  a.RT = (int)rt;
  a.tw4 = (a.tw + c4::TB_COLS - 1) / c4::TB_COLS;
  const int64_t nb = (int64_t)((rt + c4::TB_ROWS - 1) / c4::TB_ROWS) * a.tw4;
  if (nb >= (1LL << 26)) return CRB_ERR_ARG;
  a.nblocks = (int)nb;
  a.ncb = cout / c4::WG_K;
  int dev = 0;
  const int n_cu = device_cus4(&dev);
  if (n_cu <= 0) return CRB_ERR_LAUNCH;
  if (!(g_dev_attr4[dev].load(std::memory_order_acquire) & (1u << 31))) {
    CRB_HIP(hipFuncSetAttribute((const void*)winograd4c_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, c4::LDS_BYTES));
    g_dev_attr4[dev].fetch_or(1u << 31, std::memory_order_release);
  }
  const int64_t units = nb * a.ncb;
  const int64_t grid = units < n_cu ? units : n_cu;
  unsigned seq = (g_dev_seq4[dev].fetch_add(1, std::memory_order_relaxed) + 1) & 0xffffffu;
  a.seq = (grid == n_cu) ? (seq ? seq : 1) : 0;
  hipLaunchKernelGGL(winograd4c_kernel, dim3((unsigned)grid), dim3(NT), c4::LDS_BYTES, (hipStream_t)stream, a);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_conv3x3_winograd4c_nhwc(const float* x, const void* U, float* y, int N, int H, int W, int cin, int cout,
                                           const float* bias, int relu, void* stream) {
  return winograd4c_launch(x, U, y, N, H, W, cin, cout, bias, relu, stream, nullptr);
}

extern "C" int64_t crb_winograd4c_stats_slabs(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  const int64_t th = (H + 1) / 2, tw4 = ((W + 1) / 2 + c4::TB_COLS - 1) / c4::TB_COLS;
  return ((N * th + c4::TB_ROWS - 1) / c4::TB_ROWS) * tw4;       // one slab per spatial block of 32 tiles
}

extern "C" int crb_conv3x3_winograd4c_stats_nhwc(const float* x, const void* U, float* y, float* stats, int N, int H, int W, int cin,
                                                 int cout, void* stream) {
  if (!stats) return CRB_ERR_ARG;
  return winograd4c_launch(x, U, y, N, H, W, cin, cout, nullptr, 0, stream, stats);
}

static int winograd4_launch(const float* x, const void* U, float* y, int N, int H, int W, int cin, int cout, const float* bias, int relu,
                            void* stream, float* stats) {
  if (N <= 0 || H <= 0 || W <= 0) return CRB_ERR_ARG;
  if (!crb_winograd4_supported(cin, cout, H, W)) return CRB_ERR_UNSUPPORTED;
  Wino4Args a;
  a.x = x; a.U = (const unsigned char*)U; a.y = y; a.bias = bias; a.stats = stats;
  a.N = N; a.H = H; a.W = W; a.cin = cin; a.cout = cout; a.relu = relu;
  a.th = (H + 1) / 2; a.tw = (W + 1) / 2;
  const int64_t rt = (int64_t)N * a.th;
  if (rt >= (1LL << 30) || (int64_t)N * H * W * (cin > cout ? cin : cout) >= (1LL << 40)) return CRB_ERR_ARG;
  a.RT = (int)rt;
  a.tw4 = (a.tw + TB_COLS - 1) / TB_COLS;
  const int64_t nb = (int64_t)((rt + TB_ROWS - 1) / TB_ROWS) * a.tw4;
  if (nb >= (1LL << 26)) return CRB_ERR_ARG;
  a.nblocks = (int)nb;
  a.ncb = cout / WG_K;
  int mode = 0, nt = NT;
  auto kern = winograd4_kernel<0, 1>;
  int lds_bytes = LDS_BYTES_R;
#ifdef CRB_MEASURE
  mode = g_wino4_mode;
  if (g_wino4_variant == 2) {              // second form (two waves per SIMD): A/B only
    nt = NT2;
    kern = winograd4b_kernel<0>;
    lds_bytes = LDS_BYTES;
    if (mode == 1) kern = winograd4b_kernel<1>;
    if (mode == 2) kern = winograd4b_kernel<2>;
    if (mode == 3) kern = winograd4b_kernel<3>;
    if (mode == 5) kern = winograd4b_kernel<5>;
    if (mode == 6) kern = winograd4b_kernel<6>;
    if (mode == 7) kern = winograd4b_kernel<7>;
    if (mode == 8) kern = winograd4b_kernel<8>;
    if (mode == 64 + 30) kern = winograd4b_kernel<64 + 30>;       // MFMAs (+ counters, barriers) only
    if (mode == 64 + 22) kern = winograd4b_kernel<64 + 22>;       // MFMAs + operand reads
    if (mode == 64 + 31) kern = winograd4b_kernel<64 + 31>;       // counters and barriers only
    if (mode == 64 + 29) kern = winograd4b_kernel<64 + 29>;       // transform only
    mode = 32;                             // (no attribute bit: set every time)
  } else {
    const bool ur = g_wino4_variant != 3;      // variant 3: the first form with U through LDS-DMA (the round's first product kernel)
    if (!ur) lds_bytes = LDS_BYTES;
    if (mode == 0 && !ur) kern = winograd4_kernel<0, 0>;
    if (mode == 1) kern = ur ? winograd4_kernel<1, 1> : winograd4_kernel<1, 0>;
    if (mode == 2) kern = ur ? winograd4_kernel<2, 1> : winograd4_kernel<2, 0>;
    if (mode == 3) kern = ur ? winograd4_kernel<3, 1> : winograd4_kernel<3, 0>;
    if (mode == 4) kern = ur ? winograd4_kernel<4, 1> : winograd4_kernel<4, 0>;
    if (mode == 5) kern = ur ? winograd4_kernel<5, 1> : winograd4_kernel<5, 0>;
    if (mode == 6) kern = ur ? winograd4_kernel<6, 1> : winograd4_kernel<6, 0>;
    if (mode == 7) kern = ur ? winograd4_kernel<7, 1> : winograd4_kernel<7, 0>;
    if (mode == 8) kern = ur ? winograd4_kernel<8, 1> : winograd4_kernel<8, 0>;
    if (mode > 8) mode = 0;
    if (!ur || mode) mode = 32;            // (measurement instances: attribute set every time)
  }
#endif
  int dev = 0;
  const int n_cu = device_cus4(&dev);
  if (n_cu <= 0) return CRB_ERR_LAUNCH;
  if (mode >= 32 || !(g_dev_attr4[dev].load(std::memory_order_acquire) & (1u << mode))) {
    CRB_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    if (mode < 32) g_dev_attr4[dev].fetch_or(1u << mode, std::memory_order_release);
  }
  const int64_t units = nb * a.ncb;
  const int64_t grid = units < n_cu ? units : n_cu;
  unsigned seq = (g_dev_seq4[dev].fetch_add(1, std::memory_order_relaxed) + 1) & 0xffffffu;
  a.seq = (grid == n_cu) ? (seq ? seq : 1) : 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(nt), lds_bytes, (hipStream_t)stream, a);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_conv3x3_winograd4_nhwc(const float* x, const void* U, float* y, int N, int H, int W, int cin, int cout,
                                          const float* bias, int relu, void* stream) {
  return winograd4_launch(x, U, y, N, H, W, cin, cout, bias, relu, stream, nullptr);
}

extern "C" int64_t crb_winograd4_stats_slabs(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  const int64_t th = (H + 1) / 2, tw4 = ((W + 1) / 2 + TB_COLS - 1) / TB_COLS;
  return (g_wino4_variant == 2 ? 4 : 2) * ((N * th + TB_ROWS - 1) / TB_ROWS) * tw4;       // (spatial block, tile half[, output row: second form])
}

extern "C" int crb_conv3x3_winograd4_stats_nhwc(const float* x, const void* U, float* y, float* stats, int N, int H, int W, int cin,
                                                int cout, void* stream) {
  if (!stats) return CRB_ERR_ARG;
  return winograd4_launch(x, U, y, N, H, W, cin, cout, nullptr, 0, stream, stats);
}
