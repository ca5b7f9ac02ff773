// Sparse-conv "rulebook" building blocks for gfx950 (row a4 of SURVEY §8): the site hash, and the one-table-at-a-time row
// orders (mask sort, tile order) kept for A/B runs. What a training / scoring step runs is rulebook_plan.hip.
//
// Replaces the indice-pair generation inside spconv.pytorch SubMConv3d / SparseConv3d as configured by
// the reference at pcdet/models/backbones_3d/spconv_backbone.py:77-117 (third-party spconv-cu113 v2.1.21,
// absent from the reference tree; semantics restated in SURVEY Appendix A).
//
// Data layout in HBM (all int32, device):
//   coords  (N,4)   [b,z,y,x]
//   nbr     (N_out,K)  output-stationary table: nbr[i][o] = input row feeding output i through kernel
//                      offset o = (kz*KH+ky)*KW+kx, or -1
//   nbr_t   (N_in,K)   the transposed table (for dgrad of strided convs): nbr_t[j][o] = output row or -1
//   pairs   pair_in/pair_out (P) sorted by (o, out row) + pair_start (K+1): the classic rulebook, used by wgrad
// Site -> row lookup is an open-addressing hash on the 64-bit linear site index. The output set of a
// strided conv is de-duplicated AND ordered through a bitmap over the output volume (rank = popcount
// prefix), which makes the output row order canonical: ascending linear index (b,z,y,x).
#include "crb_common.h"
#include <hipcub/hipcub.hpp>
#include "../../include/crb_hip.h"

namespace {

struct Shape3 { int d, h, w; };

__device__ __forceinline__ int64_t lin_index(int b, int z, int y, int x, Shape3 s) {
  return (((int64_t)b * s.d + z) * s.h + y) * (int64_t)s.w + x;
}

__global__ __launch_bounds__(256) void hash_build_kernel(const int* __restrict__ coords, int n, Shape3 s,
                                                         long long* __restrict__ hkeys, int* __restrict__ hvals,
                                                         uint32_t hmask) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int4 c = *reinterpret_cast<const int4*>(coords + (int64_t)i * 4);
  uint32_t slot = crb_ghash_insert(hkeys, hmask, lin_index(c.x, c.y, c.z, c.w, s));
  // duplicate coordinates: the smallest row wins (deterministic)
  if (slot != 0xffffffffu) atomicMin(&hvals[slot], i);
}

// Sort the rows of every chunk of SORT_CHUNK consecutive rows by their neighbour mask, descending (stable: ties keep row
// order).
// One 1024-thread workgroup per chunk; bitonic network on 64-bit (mask << 32 | local index) keys in LDS.

// Sort key of a row: its mask with the bits re-ranked so that the RAREST offsets are the most significant. A 16-row tile pays
// a full MFMA pass for an offset as soon as ONE of its rows has it, so what has to be clustered is the rare offsets; in plain
// numeric order only the corners of the +z plane of a 3x3x3 kernel were on top. rank_bits 2 (default): bits ranked by their
// frequency inside the chunk (works for every kernel size and for transposed / strided tables); 1: by geometry (8 corners, 12
// edges, 6 faces, centre of a 3x3x3 kernel); 0: numeric. Any bijection of the bits keeps equal masks adjacent. Measured on the
// SECOND bs=16 tables (tools/ab_sort_key.py): tile fill 0.823 -> 0.849 (L3), 0.769 -> 0.830 (L4); 64x64 forward 153.6 -> 147.1
// us (L3), 115.3 -> 100.7 us (L4); 32x32 55.5 -> 52.8 us.
struct SortBits { unsigned char pos[32]; };

template <int SORT_CHUNK>
__global__ __launch_bounds__(1024) void mask_sort_chunks_kernel(const int* __restrict__ mask, int n, int* __restrict__ perm,
                                                                SortBits sb, int rank_bits) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long key[];      // SORT_CHUNK keys
  __shared__ int hist[32];
  __shared__ unsigned char lpos[32];
  const int base = blockIdx.x * SORT_CHUNK;
  if (rank_bits == 2) {                               // rank the bits by how rare they are IN THIS CHUNK
    if (threadIdx.x < 32) hist[threadIdx.x] = 0;
    __syncthreads();
    int cnt = 0;                                       // lane b of a wave accumulates the wave's count of bit b
    for (int t = threadIdx.x; t < SORT_CHUNK; t += 1024) {
      const int i = base + t;
      const unsigned m = (i < n) ? (unsigned)mask[i] : 0u;
      for (int b = 0; b < 32; ++b) {
        const int c = __popcll(__ballot((m >> b) & 1u));
        if ((int)(threadIdx.x & 63) == b) cnt += c;
      }
    }
    if ((threadIdx.x & 63) < 32 && cnt) atomicAdd(&hist[threadIdx.x & 63], cnt);
    __syncthreads();
    if (threadIdx.x < 32) {                           // position = number of bits that are more frequent (ties: lower index)
      const int mine = hist[threadIdx.x];
      int p = 0;
      for (int b = 0; b < 32; ++b) p += (hist[b] > mine) || (hist[b] == mine && b < (int)threadIdx.x);
      lpos[threadIdx.x] = (unsigned char)p;
    }
    __syncthreads();
  }
  for (int t = threadIdx.x; t < SORT_CHUNK; t += 1024) {
    const int i = base + t;
    // descending order: the rows with the most (rare) neighbours (longest-running tiles) come first inside every chunk, so
    // the last tiles a launch dispatches are light ones (no heavy-tile tail); equal masks stay adjacent either way
    unsigned m = (i < n) ? (unsigned)mask[i] : 0u;
    if (rank_bits) {
      unsigned r = 0;
      for (int b = 0; b < 32; ++b) r |= ((m >> b) & 1u) << (rank_bits == 2 ? lpos[b] : sb.pos[b]);
      m = r;
    }
    key[t] = (i < n) ? (((unsigned long long)(~m) << 32) | (unsigned)t) : ~0ULL;
  }
  __syncthreads();
  for (int k = 2; k <= SORT_CHUNK; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < SORT_CHUNK / 2; t += 1024) {
        // t-th compare-exchange of this stage: partners (i, i^j) with i the one whose bit j is clear
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int p = i | j;
        const bool up = (i & k) == 0;
        const unsigned long long a = key[i], b = key[p];
        if ((a > b) == up) { key[i] = b; key[p] = a; }
      }
      __syncthreads();
    }
  for (int t = threadIdx.x; t < SORT_CHUNK; t += 1024) {
    const int i = base + t;
    if (i < n) perm[i] = base + (int)(key[t] & 0xffffffffULL);
  }
}

// Sort keys for chunks of ANY size (crb_mask_sort_rows): one 1024-thread workgroup per chunk ranks the bits by their frequency
// inside the chunk exactly like mask_sort_chunks_kernel and writes key = chunk << 32 | ~ranked mask, value = row; a device
// radix sort (hipCUB, stable) of the (key, row) pairs then gives the same order the LDS bitonic sort gives for 4,096-row
// chunks, for chunks that do not fit one workgroup's LDS.
__global__ __launch_bounds__(1024) void mask_rank_keys_kernel(const int* __restrict__ mask, int n, int chunk_rows,
                                                              SortBits sb, int rank_bits,
                                                              unsigned long long* __restrict__ keys, int* __restrict__ rows) {
  __shared__ int hist[32];
  __shared__ unsigned char lpos[32];
  const int base = blockIdx.x * chunk_rows;
  const int end = min(n, base + chunk_rows);
  if (rank_bits == 2) {
    if (threadIdx.x < 32) hist[threadIdx.x] = 0;
    __syncthreads();
    int cnt = 0;
    for (int t0 = base; t0 < end; t0 += 1024) {      // whole waves stay in the loop: the ballot needs every lane
      const int i = t0 + threadIdx.x;
      const unsigned m = (i < end) ? (unsigned)mask[i] : 0u;
      for (int b = 0; b < 32; ++b) {
        const int c = __popcll(__ballot((m >> b) & 1u));
        if ((int)(threadIdx.x & 63) == b) cnt += c;
      }
    }
    if ((threadIdx.x & 63) < 32 && cnt) atomicAdd(&hist[threadIdx.x & 63], cnt);
    __syncthreads();
    if (threadIdx.x < 32) {
      const int mine = hist[threadIdx.x];
      int p = 0;
      for (int b = 0; b < 32; ++b) p += (hist[b] > mine) || (hist[b] == mine && b < (int)threadIdx.x);
      lpos[threadIdx.x] = (unsigned char)p;
    }
    __syncthreads();
  }
  for (int i = base + threadIdx.x; i < end; i += 1024) {
    unsigned m = (unsigned)mask[i];
    if (rank_bits) {
      unsigned r = 0;
      for (int b = 0; b < 32; ++b) r |= ((m >> b) & 1u) << (rank_bits == 2 ? lpos[b] : sb.pos[b]);
      m = r;
    }
    keys[i] = ((unsigned long long)blockIdx.x << 32) | (unsigned long long)(~m);
    rows[i] = i;
  }
}

// ---- heaviest-tile-first order of the 64-row tiles of a mask-sorted table -------------------------------------------
// A gather-GEMM workgroup runs one phase per kernel offset present in ANY of its 64 rows (1..27 phases), so tile run times
// differ by more than an order of magnitude; dispatched in table order the last tiles of a launch can be 27-phase ones and
// most CUs idle behind them. The table is cut into 8 contiguous ranges of tiles, one per XCD (workgroup b runs on XCD b % 8
// and takes tile (b % 8) * per + b / 8, see xcd_remap in sparse_conv.hip): a range keeps its rows — and with them the L2
// footprint of its gathers — on one XCD, and INSIDE each range the tiles are ranked heaviest first by popcount of the OR of
// their row masks (33-bin counting sort per range, two tiny launches). The order among equally heavy tiles is arbitrary
// (atomics) — results do not depend on it, every tile writes its own rows.
constexpr int LPT_RANGES = 8;

__global__ __launch_bounds__(256) void tile_weight_kernel(const int* __restrict__ mask, const int* __restrict__ perm,
                                                          int ntiles, int per, int* __restrict__ weight,
                                                          int* __restrict__ hist) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= ntiles) return;
  const int lane = threadIdx.x & 63;
  const unsigned m = (unsigned)mask[perm[(int64_t)t * 64 + lane]];
  int w = 0;
  for (int b = 0; b < 32; ++b) w += __ballot((m >> b) & 1u) != 0ULL;      // bits present in any of the 64 rows
  if (lane == 0) {
    weight[t] = w;
    atomicAdd(&hist[(t / per) * 64 + w], 1);
  }
}

__global__ __launch_bounds__(256) void tile_place_kernel(const int* __restrict__ weight, const int* __restrict__ hist,
                                                         int* __restrict__ cursor, int ntiles, int per,
                                                         int* __restrict__ order) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= ntiles) return;
  const int w = weight[t], r = t / per;
  int base = r * per;
  for (int b = 32; b > w; --b) base += hist[r * 64 + b];
  order[base + atomicAdd(&cursor[r * 64 + w], 1)] = t;
}

__global__ __launch_bounds__(256) void tile_perm_kernel(const int* __restrict__ perm, const int* __restrict__ order,
                                                        int n, int ntiles, int* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int t = i >> 6;
  out[i] = t < ntiles ? perm[(order[t] << 6) + (i & 63)] : perm[i];
}

}  // namespace

static SortBits geometric_sort_bits();

extern "C" int crb_mask_sort_chunk_rows(void) { return 4096; }

CRB_KNOB g_sort_rank_bits = 2;    // sort key: 2 = bits ranked rarest first inside the chunk (default), 1 = by 3x3x3 geometry, 0 = numeric
#ifdef CRB_MEASURE
extern "C" int crb_mask_sort_set_rank_bits(int mode) { g_sort_rank_bits = (mode >= 0 && mode <= 2) ? mode : 2; return CRB_OK; }
#endif

// perm (n) i32: row permutation that sorts every chunk of crb_mask_sort_chunk_rows() consecutive rows by mask (stable)
extern "C" int crb_mask_sort_chunks(const int32_t* mask, int64_t n, int32_t* perm, void* stream) {
  if (n < 0) return CRB_ERR_ARG;
  if (n == 0) return CRB_OK;
  const SortBits sb = geometric_sort_bits();
  hipLaunchKernelGGL(mask_sort_chunks_kernel<4096>, dim3(crb_cdiv(n, 4096)), dim3(1024), 4096 * 8, (hipStream_t)stream, mask,
                     (int)n, perm, sb, g_sort_rank_bits);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

static SortBits geometric_sort_bits() {
  SortBits sb;
  int next[4] = {0, 1, 7, 19};                         // first bit of the class with 0 / 1 / 2 / 3 non-zero coordinates
  for (int o = 0; o < 32; ++o) {
    if (o >= 27) { sb.pos[o] = (unsigned char)o; continue; }
    const int nz = (o / 9 != 1) + ((o / 3) % 3 != 1) + (o % 3 != 1);
    sb.pos[o] = (unsigned char)next[nz]++;
  }
  return sb;
}

static size_t radix_temp_bytes(int64_t n) {
  size_t b = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                           (const int*)nullptr, (int*)nullptr, (int)n, 0, 64, (hipStream_t)0);
  return b;
}

extern "C" int64_t crb_mask_sort_rows_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  return crb_align_up(8 * n, 256) * 2 + crb_align_up(4 * n, 256) + crb_align_up((int64_t)radix_temp_bytes(n), 256) + 256;
}

// perm (n) i32: the order crb_mask_sort_chunks produces, for chunks of chunk_rows rows (any multiple of 1024)
extern "C" int crb_mask_sort_rows(const int32_t* mask, int64_t n, int chunk_rows, int32_t* perm, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
  if (n < 0 || n >= (1LL << 31) || chunk_rows < 1024 || chunk_rows % 1024) return CRB_ERR_ARG;
  if (n == 0) return CRB_OK;
  if (!workspace || workspace_bytes < crb_mask_sort_rows_workspace_bytes(n)) return CRB_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  CrbArena ar(workspace, (size_t)workspace_bytes);
  unsigned long long* kin = ar.take<unsigned long long>(n);
  unsigned long long* kout = ar.take<unsigned long long>(n);
  int* vin = ar.take<int>(n);
  size_t tb = radix_temp_bytes(n);
  char* temp = ar.take<char>((int64_t)tb);
  if (!ar.ok) return CRB_ERR_WORKSPACE;
  const int chunks = crb_cdiv(n, chunk_rows);
  hipLaunchKernelGGL(mask_rank_keys_kernel, dim3(chunks), dim3(1024), 0, st, mask, (int)n, chunk_rows, geometric_sort_bits(),
                     g_sort_rank_bits, kin, vin);
  int chunk_bits = 1;
  while ((1 << chunk_bits) < chunks) ++chunk_bits;
  if (hipcub::DeviceRadixSort::SortPairs(temp, tb, kin, kout, vin, perm, (int)n, 0, 32 + chunk_bits, st) != hipSuccess)
    return CRB_ERR_LAUNCH;
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int64_t crb_hash_capacity_for(int64_t n) { return crb_hash_capacity(n); }

extern "C" int crb_sparse_hash_build(const int32_t* coords, int64_t n, const int32_t* shape_dhw,
                                     int64_t* hkeys, int32_t* hvals, int64_t capacity, void* stream) {
  if (n < 0 || capacity < 2 * n || (capacity & (capacity - 1))) return CRB_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  CRB_HIP(hipMemsetAsync(hkeys, 0xff, (size_t)capacity * 8, st));
  CRB_HIP(hipMemsetAsync(hvals, 0x7f, (size_t)capacity * 4, st));
  if (n == 0) return CRB_OK;
  Shape3 s{shape_dhw[0], shape_dhw[1], shape_dhw[2]};
  hipLaunchKernelGGL(hash_build_kernel, dim3(crb_cdiv(n, 256)), dim3(256), 0, st, coords, (int)n, s,
                     (long long*)hkeys, hvals, (uint32_t)(capacity - 1));
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int64_t crb_tile_lpt_workspace_bytes(int64_t n) { return (2 * LPT_RANGES * 64 + 2 * (n / 64 + 1)) * 4; }

extern "C" int crb_tile_lpt_ranges(void) { return LPT_RANGES; }

// perm_out = perm_in with the full 64-row tiles of each of the crb_tile_lpt_ranges() contiguous tile ranges re-ordered
// heaviest first (weight = number of kernel offsets present in any row of the tile); the trailing partial tile stays last.
// Range length = ceil(ceil(n/64) / ranges) tiles — the split xcd_remap uses. mask (n) is indexed by ORIGINAL row.
extern "C" int crb_tile_lpt_perm(const int32_t* mask, const int32_t* perm_in, int64_t n, int32_t* perm_out,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
  if (n < 0 || n >= (1LL << 31)) return CRB_ERR_ARG;
  if (n == 0) return CRB_OK;
  if (!workspace || workspace_bytes < crb_tile_lpt_workspace_bytes(n)) return CRB_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int ntiles = (int)(n / 64);                               // full tiles
  const int per = (int)((crb_cdiv(n, 64) + LPT_RANGES - 1) / LPT_RANGES);
  int* hist = (int*)workspace;                                    // [range][64 bins] (0..32 used)
  int* cursor = hist + LPT_RANGES * 64;
  int* weight = cursor + LPT_RANGES * 64;
  int* order = weight + ntiles + 1;
  CRB_HIP(hipMemsetAsync(hist, 0, 2 * LPT_RANGES * 64 * 4, st));
  if (ntiles > 0) {
    hipLaunchKernelGGL(tile_weight_kernel, dim3(crb_cdiv(ntiles, 4)), dim3(256), 0, st, mask, perm_in, ntiles, per, weight,
                       hist);
    hipLaunchKernelGGL(tile_place_kernel, dim3(crb_cdiv(ntiles, 256)), dim3(256), 0, st, weight, hist, cursor, ntiles, per,
                       order);
  }
  hipLaunchKernelGGL(tile_perm_kernel, dim3(crb_cdiv(n, 256)), dim3(256), 0, st, perm_in, order, (int)n, ntiles, perm_out);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}
