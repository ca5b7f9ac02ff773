// Internal helpers shared by the gfx950 kernels of libcrbhip.so (not part of the C-ABI).
// Wave = 64 lanes everywhere; no CUDA/dual-path code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CRB_OK 0
#define CRB_ERR_ARG (-1)
#define CRB_ERR_WORKSPACE (-2)
#define CRB_ERR_LAUNCH (-3)
#define CRB_ERR_UNSUPPORTED (-4)

#define CRB_WAVE 64

#define CRB_CHECK_LAUNCH()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return CRB_ERR_LAUNCH;  \
  } while (0)

#define CRB_HIP(call)                              \
  do {                                             \
    hipError_t e__ = (call);                       \
    if (e__ != hipSuccess) return CRB_ERR_LAUNCH;  \
  } while (0)

static inline int64_t crb_align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
static inline int crb_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Bump allocator over a caller-owned workspace. Every carve is 256-B aligned.
struct CrbArena {
  char* base;
  size_t cap;
  size_t off;
  bool ok;
  CrbArena(void* p, size_t n) : base((char*)p), cap(n), off(0), ok(true) {}
  template <typename T>
  T* take(int64_t count) {
    size_t bytes = (size_t)crb_align_up((int64_t)(count * sizeof(T)), 256);
    if (base == nullptr || off + bytes > cap) { ok = false; off += bytes; return nullptr; }
    T* r = (T*)(base + off);
    off += bytes;
    return r;
  }
};

// ---------------------------------------------------------------------------------------------
// wave / block primitives
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int crb_lane() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ int crb_wave_incl_scan(int v) {
  const int lane = crb_lane();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

__device__ __forceinline__ float crb_wave_sum(float v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// Block-wide exclusive scan of one int per thread (blockDim.x == 256). Returns the exclusive
// prefix; *total receives the block sum. `sh` must hold >= 4 ints.
__device__ __forceinline__ int crb_block_excl_scan_256(int v, int* sh, int* total) {
  const int lane = crb_lane();
  const int wave = (int)(threadIdx.x >> 6);
  int inc = crb_wave_incl_scan(v);
  if (lane == 63) sh[wave] = inc;
  __syncthreads();
  int woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    int s = sh[w];
    if (w < wave) woff += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return woff + inc - v;
}

// ---------------------------------------------------------------------------------------------
// Device-wide exclusive scan (int32) as three launches: tile sums -> scan of sums -> apply.
// F maps an element index to its int value. SCAN_TILE elements per 256-thread block.
// ---------------------------------------------------------------------------------------------
#define CRB_SCAN_ITEMS 8
#define CRB_SCAN_TILE (256 * CRB_SCAN_ITEMS)

template <typename F>
__global__ __launch_bounds__(256) void crb_scan_tile_sums(F f, int64_t n, int* __restrict__ tile_sums) {
  __shared__ int sh[4];
  const int64_t base = (int64_t)blockIdx.x * CRB_SCAN_TILE + (int64_t)threadIdx.x * CRB_SCAN_ITEMS;
  int s = 0;
#pragma unroll
  for (int k = 0; k < CRB_SCAN_ITEMS; ++k) {
    int64_t i = base + k;
    if (i < n) s += f(i);
  }
  int tot;
  crb_block_excl_scan_256(s, sh, &tot);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// one block; scans `m` tile sums in place (exclusive) and writes the grand total to *total_out.
static __global__ __launch_bounds__(256) void crb_scan_of_sums(int* __restrict__ tile_sums, int m, int* __restrict__ total_out) {
  __shared__ int sh[4];
  int carry = 0;
  for (int base = 0; base < m; base += 256) {
    int i = base + (int)threadIdx.x;
    int v = (i < m) ? tile_sums[i] : 0;
    int tot;
    int ex = crb_block_excl_scan_256(v, sh, &tot);
    if (i < m) tile_sums[i] = carry + ex;
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry;
}

template <typename F, typename W>
__global__ __launch_bounds__(256) void crb_scan_apply(F f, W w, int64_t n, const int* __restrict__ tile_sums) {
  __shared__ int sh[4];
  const int64_t base = (int64_t)blockIdx.x * CRB_SCAN_TILE + (int64_t)threadIdx.x * CRB_SCAN_ITEMS;
  int v[CRB_SCAN_ITEMS];
  int s = 0;
#pragma unroll
  for (int k = 0; k < CRB_SCAN_ITEMS; ++k) {
    int64_t i = base + k;
    v[k] = (i < n) ? f(i) : 0;
    s += v[k];
  }
  int tot;
  int ex = crb_block_excl_scan_256(s, sh, &tot) + tile_sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < CRB_SCAN_ITEMS; ++k) {
    int64_t i = base + k;
    if (i < n) w(i, ex, v[k]);
    ex += v[k];
  }
}

// Host driver. `tile_sums` must hold crb_scan_num_tiles(n) ints. total_out may be null.
static inline int crb_scan_num_tiles(int64_t n) { return n <= 0 ? 1 : (int)((n + CRB_SCAN_TILE - 1) / CRB_SCAN_TILE); }

template <typename F, typename W>
static inline int crb_device_excl_scan(F f, W w, int64_t n, int* tile_sums, int* total_out, hipStream_t st) {
  if (n <= 0) {
    if (total_out) { if (hipMemsetAsync(total_out, 0, sizeof(int), st) != hipSuccess) return CRB_ERR_LAUNCH; }
    return CRB_OK;
  }
  const int tiles = crb_scan_num_tiles(n);
  hipLaunchKernelGGL((crb_scan_tile_sums<F>), dim3(tiles), dim3(256), 0, st, f, n, tile_sums);
  hipLaunchKernelGGL(crb_scan_of_sums, dim3(1), dim3(256), 0, st, tile_sums, tiles, total_out);
  hipLaunchKernelGGL((crb_scan_apply<F, W>), dim3(tiles), dim3(256), 0, st, f, w, n, tile_sums);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

// ---------------------------------------------------------------------------------------------
// 64-bit key hash (open addressing, linear probing). EMPTY = -1.
// ---------------------------------------------------------------------------------------------
#define CRB_HASH_EMPTY (-1LL)

__device__ __forceinline__ uint32_t crb_hash64(int64_t k) {
  uint64_t x = (uint64_t)k;
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return (uint32_t)x;
}

// insert key; returns slot. `mask` = capacity-1 (capacity power of two, > number of keys).
__device__ __forceinline__ uint32_t crb_hash_insert(long long* __restrict__ keys, uint32_t mask, int64_t key) {
  uint32_t slot = crb_hash64(key) & mask;
  while (true) {
    long long prev = (long long)atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)CRB_HASH_EMPTY,
                                          (unsigned long long)key);
    if (prev == CRB_HASH_EMPTY || prev == (long long)key) return slot;
    slot = (slot + 1) & mask;
  }
}

// lookup; returns slot or 0xffffffff when absent. Table must be quiescent (built by an earlier launch).
__device__ __forceinline__ uint32_t crb_hash_find(const long long* __restrict__ keys, uint32_t mask, int64_t key) {
  uint32_t slot = crb_hash64(key) & mask;
  while (true) {
    long long k = keys[slot];
    if (k == (long long)key) return slot;
    if (k == CRB_HASH_EMPTY) return 0xffffffffu;
    slot = (slot + 1) & mask;
  }
}

// The same table with x-GROUPED slots (site hash of the sparse-conv rulebooks): slot = group(key / 8) * 8 + key % 8 and a
// collision moves on by whole groups — eight interleaved open-addressing tables, one per key % 8, that share the group hash.
// Keys are linear site indices with x fastest, so the 3 x-neighbours a kernel row looks up fall into ONE 64-byte line of
// keys most of the time (a separate hash per site touched 27 lines per row: the level-1 SubM table took 115 us).
// A residue class that fills ALL of its capacity/8 slots (a wall at constant x with W % 8 == 0: every key shares key % 8)
// spills into the next residue class after one full round, and a lookup that met no empty slot in a round follows it there
// (the table is quiescent when it is searched): probing always terminates, an over-full class costs time, never a hang
// (ADVICE r03). crb_hash_capacity additionally sizes small tables so that no class can fill up at all.
__device__ __forceinline__ uint32_t crb_ghash_insert(long long* __restrict__ keys, uint32_t mask, int64_t key) {
  const uint32_t gmask = mask >> 3;
  uint32_t sub = (uint32_t)(key & 7);
  const uint32_t g0 = crb_hash64(key >> 3) & gmask;
  for (int round = 0; round < 8; ++round, sub = (sub + 1) & 7) {
    uint32_t g = g0;
    do {
      const uint32_t slot = (g << 3) | sub;
      long long prev = (long long)atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)CRB_HASH_EMPTY,
                                            (unsigned long long)key);
      if (prev == CRB_HASH_EMPTY || prev == (long long)key) return slot;
      g = (g + 1) & gmask;
    } while (g != g0);
  }
  return 0xffffffffu;          // table completely full: capacity > number of keys is the caller's contract
}

__device__ __forceinline__ uint32_t crb_ghash_find(const long long* __restrict__ keys, uint32_t mask, int64_t key) {
  const uint32_t gmask = mask >> 3;
  uint32_t sub = (uint32_t)(key & 7);
  const uint32_t g0 = crb_hash64(key >> 3) & gmask;
  for (int round = 0; round < 8; ++round, sub = (sub + 1) & 7) {
    uint32_t g = g0;
    do {
      const uint32_t slot = (g << 3) | sub;
      const long long k = keys[slot];
      if (k == (long long)key) return slot;
      if (k == CRB_HASH_EMPTY) return 0xffffffffu;
      g = (g + 1) & gmask;
    } while (g != g0);
  }
  return 0xffffffffu;
}

static inline int64_t crb_hash_capacity(int64_t n) {
  int64_t c = 1024;
  while (c < 2 * n) c <<= 1;
  // the x-grouped tables split the capacity into 8 residue classes of c / 8 slots: up to 128 Ki keys give every class room
  // for ALL keys (c / 8 > n: a one-residue input cannot fill its class; <= 16 MB of keys + 8 MB of values); above that a full
  // class spills (crb_ghash_insert)
  if (n <= (1 << 17))
    while (c < 8 * n + 8) c <<= 1;
  return c;
}

// Measurement knobs (kernel-variant selectors, builds that skip work and return WRONG results by design, per-workgroup
// timelines) exist only in libcrbhip_measure.so (-DCRB_MEASURE, include/crb_hip_measure.h, used by tools/). In the product
// library they are compile-time constants: no setter is exported and the variants are not even instantiated.
#ifdef CRB_MEASURE
#define CRB_KNOB static int
#else
#define CRB_KNOB static constexpr int
#endif

// winograd_conv4.hip: sets ITS copy of the busy-CU word (device variables are per translation unit); crb_cu_reservation calls it
int crbhip_wino4_cu_busy_set(int cus, hipStream_t stream);
