// Fused BatchNorm1d (+ReLU) over the rows of a sparse tensor (row a5 of SURVEY §8).
// Replaces the separate nn.BatchNorm1d(eps=1e-3, momentum=0.01) + nn.ReLU passes that follow every sparse conv in
// pcdet/models/backbones_3d/spconv_backbone.py:21-25,73 (stats pass, normalise pass, ReLU pass forward; ReLU backward,
// BN reduce, BN apply backward): here forward = 1 stats pass + 1 apply pass, backward = 1 reduce pass + 1 apply pass.
//   forward : mean_c, var_c (biased) over rows;  z = relu(gamma * (x - mean) * rsqrt(var + eps) + beta)
//   backward: d = dz * [z > 0]; dbeta = sum d; dgamma = sum d * xhat;  dx = gamma * invstd * (d - dbeta/N - xhat * dgamma/N)
// Layout: x (N, C) row-major, C in {16, 32, 64, 128, ...} (multiple of 4). A 256-thread workgroup is (256/TC) row lanes x TC
// channel lanes (TC = min(C,256)/4 float4 columns); per-block partials are reduced in double by a second tiny kernel
// (fixed order: deterministic).
#include "crb_common.h"
#include "../../include/crb_hip.h"

typedef float f4 __attribute__((ext_vector_type(4)));

namespace {

// rows per statistics workgroup: sized so that a launch has ~1024 workgroups = 4 per CU (a fixed 2048 rows gave 94
// workgroups for the 190k-row sparse tensors — a third of the CUs idle — and 275 for the dense BEV rows; 4096 workgroups
// made the finalize pass, which walks all partials, the most expensive BN kernel of the step), never fewer than 128 rows
static inline int bn_rows_per_block(int64_t n) {
  int64_t r = (n + 1023) / 1024;
  r = r < 128 ? 128 : r;
  return (int)r;
}

// traversal order of the four streaming passes (bit 0: statistics passes walk the rows from the end, bit 1: apply passes do);
// the partial of a row range is stored at the range's index either way, so the finalize order and every result are unchanged.
// Default 2: the apply passes walk from the end — what the statistics pass read last is what the 256 MB Infinity Cache still
// holds (tools/time_bn.py: 144 MB forward 116 -> 103 us, 288 MB 167.6 -> 165.2 us, larger tensors unchanged)
CRB_KNOB g_bn_order = 2;

// ---- finalize inside the statistics launch ("last block done" tickets) --------------------------------------------
// The per-block partials used to be summed by a launch of their own (bn_finalize_kernel: 52 launches of ~12 us per SECOND
// step). With a ticket area the statistics kernels do it themselves, in two levels and in exactly the order of
// bn_finalize_kernel (bit-identical results): block k belongs to group k % 32; the LAST block of a group to finish sums the
// group's partials k, k+32, ... in double (what one of the 32 `part` lanes of bn_finalize_kernel does), the LAST group to
// finish adds the 32 group sums in group order and writes the statistics. Visibility across workgroups / XCDs: the values
// that cross workgroups (partials, group sums) are written with agent-scope relaxed atomic stores (`sc1`, write-through) and
// read with agent-scope relaxed atomic loads (`sc1`, L1 bypassed); every thread waits for the acknowledgement of its stores
// (s_waitcnt vmcnt(0)) before the block's barrier and ticket. tickets[0] counts groups, tickets[1 + g] the blocks of group g; whoever
// finishes a counter resets it, so the area is zero again when the kernel ends.
constexpr int BN_GROUPS = 32;
struct BnFinal {
  int* tickets;        // 1 + BN_GROUPS ints, zero on entry; nullptr = separate finalize launch
  double* group;       // BN_GROUPS x 2C doubles of scratch
  float* o0;           // forward: mean      backward: dbeta
  float* o1;           // forward: var       backward: dgamma
  float* o2;           // forward: invstd
  float* running_mean;
  float* running_var;
  long long* num_batches_tracked;   // forward: += 1 by the finalizing thread (nn.BatchNorm's counter), nullable
  float momentum, eps;
  int64_t n;
  int bwd;
};

__device__ __forceinline__ void st_agent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent4(float* p, f4 v) {
  st_agent(p, v[0]);
  st_agent(p + 1, v[1]);
  st_agent(p + 2, v[2]);
  st_agent(p + 3, v[3]);
}

__device__ __forceinline__ void bn_ticket_finalize(const float* partial, int nblk, int C, int blk, const BnFinal& f) {
  __shared__ int s_role;
  // The partials were stored write-through (agent-scope relaxed atomic stores = `sc1` stores) and are read with `sc1` loads
  // (L1 bypassed): the valid "sc1 both sides" hand-off of MI355X_MICROARCH.md §inter-workgroup visibility. What that form
  // needs before the flag is that every store of every wave has been ACKNOWLEDGED: a workgroup-scope release fence emits no
  // s_waitcnt vmcnt(0) on gfx950 and s_barrier does not drain stores, so each thread drains its own queue
  // explicitly (inline asm: the compiler cannot drop it), then the barrier, then thread 0 takes the ticket.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int ngroups = nblk < BN_GROUPS ? nblk : BN_GROUPS;
  const int grp = blk % BN_GROUPS;
  const int members = (nblk - grp + BN_GROUPS - 1) / BN_GROUPS;
  if (threadIdx.x == 0)
    s_role = __hip_atomic_fetch_add(&f.tickets[1 + grp], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1 ? 1 : 0;
  __syncthreads();
  if (!s_role) return;
  const int C2 = 2 * C;
  for (int i = threadIdx.x; i < C2; i += 256) {
    double a = 0.0;
    int k = grp;
    for (; k + 7 * BN_GROUPS < nblk; k += 8 * BN_GROUPS) {      // 8 loads in flight, added in order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = __hip_atomic_load(partial + (int64_t)(k + u * BN_GROUPS) * C2 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int u = 0; u < 8; ++u) a += (double)v[u];
    }
    for (; k < nblk; k += BN_GROUPS)
      a += (double)__hip_atomic_load(partial + (int64_t)k * C2 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(f.group + (int64_t)grp * C2 + i, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the group sums (sc1 stores) are acknowledged before the group ticket
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(&f.tickets[1 + grp], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_role = __hip_atomic_fetch_add(&f.tickets[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1 ? 2 : 0;
  }
  __syncthreads();
  if (s_role != 2) return;
  for (int c = threadIdx.x; c < C; c += 256) {
    double a = 0.0, b = 0.0;
    int g = 0;
    for (; g + 7 < ngroups; g += 8) {
      double va[8], vb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        va[u] = __hip_atomic_load(f.group + (int64_t)(g + u) * C2 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        vb[u] = __hip_atomic_load(f.group + (int64_t)(g + u) * C2 + C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { a += va[u]; b += vb[u]; }
    }
    for (; g < ngroups; ++g) {
      a += __hip_atomic_load(f.group + (int64_t)g * C2 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      b += __hip_atomic_load(f.group + (int64_t)g * C2 + C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!f.bwd) {
      const double m = a / (double)f.n;
      double v = b / (double)f.n - m * m;
      if (v < 0.0) v = 0.0;
      f.o0[c] = (float)m;
      f.o1[c] = (float)v;
      f.o2[c] = (float)(1.0 / sqrt(v + (double)f.eps));
      if (f.running_mean) {
        const float unbiased = (float)(f.n > 1 ? v * ((double)f.n / (double)(f.n - 1)) : v);
        f.running_mean[c] = f.running_mean[c] * (1.f - f.momentum) + f.momentum * (float)m;
        f.running_var[c] = f.running_var[c] * (1.f - f.momentum) + f.momentum * unbiased;
      }
    } else {
      f.o0[c] = (float)a;
      f.o1[c] = (float)b;
    }
  }
  if (threadIdx.x == 0) {
    __hip_atomic_store(&f.tickets[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (f.num_batches_tracked) *f.num_batches_tracked += 1;
  }
}

// partial[(blk * 2 + which) * C + c]; which 0: sum a, 1: sum b
template <bool BWD>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int64_t n, int C, int relu, int rows_per_block,
                                                         int64_t ld_dz, float* __restrict__ partial, int reversed, BnFinal fin) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f4* red = reinterpret_cast<f4*>(smem);               // 2 * 256 f4
  const int c4n = C >> 2;                              // float4 columns
  const int tc = threadIdx.x % c4n, tr = threadIdx.x / c4n;
  const int rlanes = 256 / c4n;
  const int blk = reversed ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  const int64_t r0 = (int64_t)blk * rows_per_block;
  const int64_t r1 = min(n, r0 + rows_per_block);
  f4 a = (f4){0, 0, 0, 0}, b = (f4){0, 0, 0, 0};
  f4 mu, is, ga, be;
  if (BWD) {
    mu = *reinterpret_cast<const f4*>(mean + tc * 4);
    is = *reinterpret_cast<const f4*>(invstd + tc * 4);
    ga = *reinterpret_cast<const f4*>(gamma + tc * 4);
    be = *reinterpret_cast<const f4*>(beta + tc * 4);
  }
  if (tr < rlanes)
    for (int64_t r = r0 + tr; r < r1; r += rlanes) {
      const f4 v = *reinterpret_cast<const f4*>(x + r * C + tc * 4);
      if (!BWD) {
        a += v;
        b += v * v;
      } else {
        const f4 xh = (v - mu) * is;
        f4 d = *reinterpret_cast<const f4*>(dz + r * ld_dz + tc * 4);
        if (relu) {
          const f4 z = ga * xh + be;
#pragma unroll
          for (int k = 0; k < 4; ++k) d[k] = z[k] > 0.f ? d[k] : 0.f;
        }
        a += d;
        b += d * xh;
      }
    }
  red[threadIdx.x] = a;
  red[256 + threadIdx.x] = b;
  __syncthreads();
  if (tr == 0) {
    for (int k = 1; k < rlanes; ++k) { a += red[k * c4n + tc]; b += red[256 + k * c4n + tc]; }
    if (fin.tickets) {
      st_agent4(partial + ((int64_t)blk * 2 + 0) * C + tc * 4, a);
      st_agent4(partial + ((int64_t)blk * 2 + 1) * C + tc * 4, b);
    } else {
      *reinterpret_cast<f4*>(partial + ((int64_t)blk * 2 + 0) * C + tc * 4) = a;
      *reinterpret_cast<f4*>(partial + ((int64_t)blk * 2 + 1) * C + tc * 4) = b;
    }
  }
  if (fin.tickets) bn_ticket_finalize(partial, gridDim.x, C, blk, fin);
}

// forward finalize: mean, biased var, invstd ; backward finalize: dbeta, dgamma.
// 256 threads = 8 channels x 32 strided slices of the per-block partials, summed in double and combined through LDS in a
// fixed order (deterministic).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int nblk, int C, int64_t n,
                                                          float eps, int bwd, float* __restrict__ o0,
                                                          float* __restrict__ o1, float* __restrict__ o2,
                                                          float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, float momentum,
                                                          long long* __restrict__ num_batches_tracked = nullptr) {
  __shared__ double sa[32][8], sb[32][8];
  if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
  const int cl = threadIdx.x & 7, part = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  double a = 0.0, b = 0.0;
  if (c < C)
    for (int k = part; k < nblk; k += 32) {
      a += (double)partial[((int64_t)k * 2 + 0) * C + c];
      b += (double)partial[((int64_t)k * 2 + 1) * C + c];
    }
  sa[part][cl] = a;
  sb[part][cl] = b;
  __syncthreads();
  if (part != 0 || c >= C) return;
  for (int k = 1; k < 32; ++k) { a += sa[k][cl]; b += sb[k][cl]; }
  if (!bwd) {
    const double m = a / (double)n;
    double v = b / (double)n - m * m;
    if (v < 0.0) v = 0.0;
    o0[c] = (float)m;
    o1[c] = (float)v;
    o2[c] = (float)(1.0 / sqrt(v + (double)eps));
    if (running_mean) {                  // nn.BatchNorm: running = (1 - m) * running + m * batch (variance unbiased)
      const float unbiased = (float)(n > 1 ? v * ((double)n / (double)(n - 1)) : v);
      running_mean[c] = running_mean[c] * (1.f - momentum) + momentum * (float)m;
      running_var[c] = running_var[c] * (1.f - momentum) + momentum * unbiased;
    }
  } else {
    o0[c] = (float)a;      // dbeta
    o1[c] = (float)b;      // dgamma
  }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ z, int64_t total4,
                                                       int C, int relu, int64_t ld_z, int reversed) {
  const int64_t t = (int64_t)(reversed ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * 256 + threadIdx.x;
  if (t >= total4) return;
  const int c = (int)((t * 4) % C);
  const f4 v = *reinterpret_cast<const f4*>(x + t * 4);
  const f4 mu = *reinterpret_cast<const f4*>(mean + c), is = *reinterpret_cast<const f4*>(invstd + c);
  const f4 ga = *reinterpret_cast<const f4*>(gamma + c), be = *reinterpret_cast<const f4*>(beta + c);
  f4 o = ga * ((v - mu) * is) + be;
  if (relu) {
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = o[k] > 0.f ? o[k] : 0.f;
  }
  *reinterpret_cast<f4*>(z + ((t * 4) / C) * ld_z + c) = o;      // ld_z == C: the dense case
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ dbeta, const float* __restrict__ dgamma,
                                                           float* __restrict__ dx, int64_t total4, int C, float inv_n,
                                                           int relu, int64_t ld_dz, int reversed) {
  const int64_t t = (int64_t)(reversed ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * 256 + threadIdx.x;
  if (t >= total4) return;
  const int c = (int)((t * 4) % C);
  const f4 v = *reinterpret_cast<const f4*>(x + t * 4);
  f4 d = *reinterpret_cast<const f4*>(dz + ((t * 4) / C) * ld_dz + c);
  const f4 mu = *reinterpret_cast<const f4*>(mean + c), is = *reinterpret_cast<const f4*>(invstd + c);
  const f4 ga = *reinterpret_cast<const f4*>(gamma + c), be = *reinterpret_cast<const f4*>(beta + c);
  const f4 db = *reinterpret_cast<const f4*>(dbeta + c), dg = *reinterpret_cast<const f4*>(dgamma + c);
  const f4 xh = (v - mu) * is;
  if (relu) {
    const f4 z = ga * xh + be;
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = z[k] > 0.f ? d[k] : 0.f;
  }
  *reinterpret_cast<f4*>(dx + t * 4) = ga * is * (d - db * inv_n - xh * (dg * inv_n));
}

// ---- BatchNorm + ReLU + max over groups of `ns` consecutive rows (the tail of a set-abstraction scale in training:
// pointnet2_modules.py:96-103, BatchNorm2d -> ReLU -> max_pool2d over nsample). The normalised (groups*ns, C) matrix is
// never written: forward keeps the max and the first row attaining it, backward rebuilds the (one-hot) upstream gradient
// from them. z is formed exactly as bn_apply_kernel forms it.
__global__ __launch_bounds__(256) void bn_relu_max_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int64_t groups, int ns, int C,
                                                          float* __restrict__ zmax, int64_t ld_out, int* __restrict__ arg) {
  const int c4n = C >> 2;
  const int tc = threadIdx.x % c4n, tr = threadIdx.x / c4n;
  const int glanes = 256 / c4n;
  if (tr >= glanes) return;
  const int64_t m = (int64_t)blockIdx.x * glanes + tr;
  if (m >= groups) return;
  const int c = tc * 4;
  const f4 mu = *reinterpret_cast<const f4*>(mean + c), is = *reinterpret_cast<const f4*>(invstd + c);
  const f4 ga = *reinterpret_cast<const f4*>(gamma + c), be = *reinterpret_cast<const f4*>(beta + c);
  const float* src = x + m * ns * C + c;
  f4 best = (f4){-1.f, -1.f, -1.f, -1.f};                   // relu output is >= 0: the first row always replaces it
  int bi[4] = {0, 0, 0, 0};
  for (int s = 0; s < ns; ++s) {
    const f4 v = *reinterpret_cast<const f4*>(src + (int64_t)s * C);
    f4 o = ga * ((v - mu) * is) + be;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o[k] = o[k] > 0.f ? o[k] : 0.f;
      if (o[k] > best[k]) { best[k] = o[k]; bi[k] = s; }
    }
  }
  *reinterpret_cast<f4*>(zmax + m * ld_out + c) = best;
  int* a = arg + m * C + c;
  a[0] = bi[0]; a[1] = bi[1]; a[2] = bi[2]; a[3] = bi[3];
}

// backward sums from the groups only: d = g[m][c] where the max is positive; dbeta = sum d, dgamma = sum d * xhat(arg row)
__global__ __launch_bounds__(256) void bn_max_partial_kernel(const float* __restrict__ x, const float* __restrict__ gz,
                                                             int64_t ld_g, const int* __restrict__ arg,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             int64_t groups, int ns, int C, int groups_per_block,
                                                             float* __restrict__ partial, BnFinal fin) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f4* red = reinterpret_cast<f4*>(smem);
  const int c4n = C >> 2;
  const int tc = threadIdx.x % c4n, tr = threadIdx.x / c4n;
  const int rlanes = 256 / c4n;
  const int c = tc * 4;
  const int64_t m0 = (int64_t)blockIdx.x * groups_per_block;
  const int64_t m1 = min(groups, m0 + groups_per_block);
  f4 a = (f4){0, 0, 0, 0}, b = (f4){0, 0, 0, 0};
  const f4 mu = *reinterpret_cast<const f4*>(mean + c), is = *reinterpret_cast<const f4*>(invstd + c);
  const f4 ga = *reinterpret_cast<const f4*>(gamma + c), be = *reinterpret_cast<const f4*>(beta + c);
  if (tr < rlanes)
    for (int64_t m = m0 + tr; m < m1; m += rlanes) {
      const f4 g = *reinterpret_cast<const f4*>(gz + m * ld_g + c);
      const int* ar = arg ? arg + m * C + c : nullptr;      // arg == NULL: x holds the selected row of every group (ns = 1)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float v = x[(m * ns + (ar ? ar[k] : 0)) * C + c + k];
        const float xh = (v - mu[k]) * is[k];
        const float d = (ga[k] * xh + be[k]) > 0.f ? g[k] : 0.f;
        a[k] += d;
        b[k] += d * xh;
      }
    }
  red[threadIdx.x] = a;
  red[256 + threadIdx.x] = b;
  __syncthreads();
  if (tr == 0) {
    for (int k = 1; k < rlanes; ++k) { a += red[k * c4n + tc]; b += red[256 + k * c4n + tc]; }
    if (fin.tickets) {
      st_agent4(partial + ((int64_t)blockIdx.x * 2 + 0) * C + c, a);
      st_agent4(partial + ((int64_t)blockIdx.x * 2 + 1) * C + c, b);
    } else {
      *reinterpret_cast<f4*>(partial + ((int64_t)blockIdx.x * 2 + 0) * C + c) = a;
      *reinterpret_cast<f4*>(partial + ((int64_t)blockIdx.x * 2 + 1) * C + c) = b;
    }
  }
  if (fin.tickets) bn_ticket_finalize(partial, gridDim.x, C, blockIdx.x, fin);
}

__global__ __launch_bounds__(256) void bn_max_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ gz,
                                                               int64_t ld_g, const int* __restrict__ arg,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ dbeta,
                                                               const float* __restrict__ dgamma, float* __restrict__ dx,
                                                               int64_t total4, int ns, int C, float inv_n) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total4) return;
  // C/4 divides 256 and 256 is the block size: a thread's column only depends on threadIdx; rows fit 32 bits (host check)
  const int c4n = C >> 2;
  const int c = (int)(threadIdx.x % c4n) * 4;
  const unsigned row = (unsigned)((t * 4) >> (31 - __clz(C)));           // C = 4 * 2^k
  const unsigned m = row / (unsigned)ns;
  const int s = (int)(row - m * (unsigned)ns);
  const f4 v = *reinterpret_cast<const f4*>(x + t * 4);
  const f4 g = *reinterpret_cast<const f4*>(gz + (int64_t)m * ld_g + c);
  typedef int i4 __attribute__((ext_vector_type(4)));
  const i4 ar = *reinterpret_cast<const i4*>(arg + (int64_t)m * C + c);
  const f4 mu = *reinterpret_cast<const f4*>(mean + c), is = *reinterpret_cast<const f4*>(invstd + c);
  const f4 ga = *reinterpret_cast<const f4*>(gamma + c), be = *reinterpret_cast<const f4*>(beta + c);
  const f4 db = *reinterpret_cast<const f4*>(dbeta + c), dg = *reinterpret_cast<const f4*>(dgamma + c);
  const f4 xh = (v - mu) * is;
  const f4 z = ga * xh + be;
  f4 d;
#pragma unroll
  for (int k = 0; k < 4; ++k) d[k] = (ar[k] == s && z[k] > 0.f) ? g[k] : 0.f;
  *reinterpret_cast<f4*>(dx + t * 4) = ga * is * (d - db * inv_n - xh * (dg * inv_n));
}

// ---- a batch of frames with PER-FRAME statistics in four launches (CRB stage 2: G frames per training-mode pass, every
// BatchNorm keeps the statistics of the reference's bs=1 loop). A frame's rows are cut into the SAME blocks a single-frame
// call would use (rows per block and block count follow from the frame's own row count), the partials are summed in the
// same fixed order, the running statistics are advanced frame after frame: bit-identical to n_frames calls of
// crb_bn_relu_forward, in 4 launches instead of 3 per frame.
#define BN_MAXF 64
struct BnFrames {
  long long off[BN_MAXF + 1];      // row range of frame f: [off[f], off[f+1])
  int rpb[BN_MAXF];                // rows per statistics block of frame f
  int nblk[BN_MAXF];               // statistics blocks of frame f
  int pbase[BN_MAXF];              // first partial slot of frame f
};

__global__ __launch_bounds__(256) void bn_partial_frames_kernel(const float* __restrict__ x, int C, BnFrames fr,
                                                                float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f4* red = reinterpret_cast<f4*>(smem);
  const int f = blockIdx.y;
  if ((int)blockIdx.x >= fr.nblk[f]) return;
  const int c4n = C >> 2;
  const int tc = threadIdx.x % c4n, tr = threadIdx.x / c4n;
  const int rlanes = 256 / c4n;
  const int64_t r0 = fr.off[f] + (int64_t)blockIdx.x * fr.rpb[f];
  const int64_t r1 = min((int64_t)fr.off[f + 1], r0 + fr.rpb[f]);
  f4 a = (f4){0, 0, 0, 0}, b = (f4){0, 0, 0, 0};
  if (tr < rlanes)
    for (int64_t r = r0 + tr; r < r1; r += rlanes) {
      const f4 v = *reinterpret_cast<const f4*>(x + r * C + tc * 4);
      a += v;
      b += v * v;
    }
  red[threadIdx.x] = a;
  red[256 + threadIdx.x] = b;
  __syncthreads();
  if (tr == 0) {
    for (int k = 1; k < rlanes; ++k) { a += red[k * c4n + tc]; b += red[256 + k * c4n + tc]; }
    const int64_t slot = fr.pbase[f] + blockIdx.x;
    *reinterpret_cast<f4*>(partial + (slot * 2 + 0) * C + tc * 4) = a;
    *reinterpret_cast<f4*>(partial + (slot * 2 + 1) * C + tc * 4) = b;
  }
}

// stats[f] = {mean (C), biased var (C), invstd (C), unbiased var (C)}; same summation order and roundings as bn_finalize_kernel
__global__ __launch_bounds__(256) void bn_finalize_frames_kernel(const float* __restrict__ partial, int C, float eps,
                                                                 BnFrames fr, float* __restrict__ stats) {
  __shared__ double sa[32][8], sb[32][8];
  const int f = blockIdx.y;
  const int64_t n = fr.off[f + 1] - fr.off[f];
  if (n <= 0) return;
  const int nblk = fr.nblk[f];
  const float* p = partial + (int64_t)fr.pbase[f] * 2 * C;
  const int cl = threadIdx.x & 7, part = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  double a = 0.0, b = 0.0;
  if (c < C)
    for (int k = part; k < nblk; k += 32) {
      a += (double)p[((int64_t)k * 2 + 0) * C + c];
      b += (double)p[((int64_t)k * 2 + 1) * C + c];
    }
  sa[part][cl] = a;
  sb[part][cl] = b;
  __syncthreads();
  if (part != 0 || c >= C) return;
  for (int k = 1; k < 32; ++k) { a += sa[k][cl]; b += sb[k][cl]; }
  const double m = a / (double)n;
  double v = b / (double)n - m * m;
  if (v < 0.0) v = 0.0;
  float* o = stats + (int64_t)f * 4 * C;
  o[c] = (float)m;
  o[C + c] = (float)v;
  o[2 * C + c] = (float)(1.0 / sqrt(v + (double)eps));
  o[3 * C + c] = (float)(n > 1 ? v * ((double)n / (double)(n - 1)) : v);
}

// running statistics advanced once per frame, in frame order (what n_frames single-frame calls do)
__global__ __launch_bounds__(256) void bn_running_frames_kernel(const float* __restrict__ stats, int C, int n_frames,
                                                                BnFrames fr, float momentum,
                                                                float* __restrict__ running_mean,
                                                                float* __restrict__ running_var) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float rm = running_mean[c], rv = running_var[c];
  for (int f = 0; f < n_frames; ++f) {
    const int64_t n = fr.off[f + 1] - fr.off[f];
    if (n <= 0) continue;
    const float m = stats[(int64_t)f * 4 * C + c];
    const float unbiased = stats[(int64_t)f * 4 * C + 3 * C + c];
    rm = rm * (1.f - momentum) + momentum * m;
    rv = rv * (1.f - momentum) + momentum * unbiased;
  }
  running_mean[c] = rm;
  running_var[c] = rv;
}

__global__ __launch_bounds__(256) void bn_apply_frames_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ z, int C, int relu, int64_t ld_z,
                                                              BnFrames fr) {
  const int f = blockIdx.y;
  const int64_t rows = fr.off[f + 1] - fr.off[f];
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;          // float4 index inside the frame
  if (t * 4 >= rows * C) return;
  const int c = (int)((t * 4) % C);
  const int64_t row = fr.off[f] + (t * 4) / C;
  const float* st = stats + (int64_t)f * 4 * C;
  const f4 v = *reinterpret_cast<const f4*>(x + row * C + c);
  const f4 mu = *reinterpret_cast<const f4*>(st + c), is = *reinterpret_cast<const f4*>(st + 2 * C + c);
  const f4 ga = *reinterpret_cast<const f4*>(gamma + c), be = *reinterpret_cast<const f4*>(beta + c);
  f4 o = ga * ((v - mu) * is) + be;
  if (relu) {
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = o[k] > 0.f ? o[k] : 0.f;
  }
  *reinterpret_cast<f4*>(z + row * ld_z + c) = o;
}

// bn_relu_max_kernel with the statistics of the group's frame (groups_per_frame consecutive groups per frame)
__global__ __launch_bounds__(256) void bn_relu_max_frames_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 int64_t groups, int64_t groups_per_frame, int ns, int C,
                                                                 float* __restrict__ zmax, int64_t ld_out,
                                                                 int* __restrict__ arg) {
  const int c4n = C >> 2;
  const int tc = threadIdx.x % c4n, tr = threadIdx.x / c4n;
  const int glanes = 256 / c4n;
  if (tr >= glanes) return;
  const int64_t m = (int64_t)blockIdx.x * glanes + tr;
  if (m >= groups) return;
  const int c = tc * 4;
  const float* st = stats + (m / groups_per_frame) * 4 * C;
  const f4 mu = *reinterpret_cast<const f4*>(st + c), is = *reinterpret_cast<const f4*>(st + 2 * C + c);
  const f4 ga = *reinterpret_cast<const f4*>(gamma + c), be = *reinterpret_cast<const f4*>(beta + c);
  const float* src = x + m * ns * C + c;
  f4 best = (f4){-1.f, -1.f, -1.f, -1.f};
  int bi[4] = {0, 0, 0, 0};
  for (int s = 0; s < ns; ++s) {
    const f4 v = *reinterpret_cast<const f4*>(src + (int64_t)s * C);
    f4 o = ga * ((v - mu) * is) + be;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o[k] = o[k] > 0.f ? o[k] : 0.f;
      if (o[k] > best[k]) { best[k] = o[k]; bi[k] = s; }
    }
  }
  *reinterpret_cast<f4*>(zmax + m * ld_out + c) = best;
  int* a = arg + m * C + c;
  a[0] = bi[0]; a[1] = bi[1]; a[2] = bi[2]; a[3] = bi[3];
}

}  // namespace

static inline int bn_blocks(int64_t n) { return crb_cdiv(n, bn_rows_per_block(n)); }

// per-block partials (256-B aligned) + the group sums of the ticket finalize
static inline int64_t bn_partial_bytes(int64_t n, int C) { return crb_align_up((int64_t)bn_blocks(n) * 2 * C * 4, 256); }
extern "C" int64_t crb_bn_workspace_bytes(int64_t n, int C) {
  return bn_partial_bytes(n, C) + (int64_t)BN_GROUPS * 2 * C * 8 + 256;
}
extern "C" int crb_bn_ticket_ints(void) { return 1 + BN_GROUPS; }

static inline BnFinal bn_final(int32_t* tickets, void* workspace, int64_t n, int C, int bwd, float* o0, float* o1, float* o2,
                               float* running_mean, float* running_var, float momentum, float eps,
                               int64_t* num_batches_tracked = nullptr) {
  BnFinal f;
  f.tickets = tickets;
  f.group = reinterpret_cast<double*>(static_cast<char*>(workspace) + bn_partial_bytes(n, C));
  f.o0 = o0;
  f.o1 = o1;
  f.o2 = o2;
  f.running_mean = running_mean;
  f.running_var = running_var;
  f.num_batches_tracked = reinterpret_cast<long long*>(num_batches_tracked);
  f.momentum = momentum;
  f.eps = eps;
  f.n = n;
  f.bwd = bwd;
  return f;
}

// training forward. mean/var/invstd (C) out. z may alias x? no: x is kept for backward.
extern "C" int crb_bn_relu_forward(const float* x, int64_t n, int C, const float* gamma, const float* beta, float eps,
                                   int relu, float* z, int64_t z_row_stride, float* mean, float* var, float* invstd,
                                   float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                                   void* workspace, int64_t workspace_bytes, int32_t* tickets, void* stream) {
  const int64_t ld_z = z_row_stride > 0 ? z_row_stride : C;
  if (ld_z < C || (ld_z & 3)) return CRB_ERR_ARG;
  if (n <= 0 || C <= 0 || (C & 3) || C > 1024 || (256 % (C >> 2) && (C >> 2) < 256)) return CRB_ERR_ARG;
  if (workspace_bytes < crb_bn_workspace_bytes(n, C) - 256 || !workspace) return CRB_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = bn_blocks(n);
  float* partial = (float*)workspace;
  hipLaunchKernelGGL(bn_partial_kernel<false>, dim3(nblk), dim3(256), 2 * 256 * 16, st, x, nullptr, nullptr, nullptr,
                     nullptr, nullptr, n, C, relu, bn_rows_per_block(n), (int64_t)C, partial, g_bn_order & 1,
                     bn_final(tickets, workspace, n, C, 0, mean, var, invstd, running_mean, running_var, momentum, eps,
                              num_batches_tracked));
  if (!tickets)
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(crb_cdiv(C, 8)), dim3(256), 0, st, partial, nblk, C, n, eps, 0, mean, var,
                       invstd, running_mean, running_var, momentum, reinterpret_cast<long long*>(num_batches_tracked));
  const int64_t total4 = n * C / 4;
  if (z)                                       // z == NULL: statistics (and running statistics) only
    hipLaunchKernelGGL(bn_apply_kernel, dim3(crb_cdiv(total4, 256)), dim3(256), 0, st, x, mean, invstd, gamma, beta, z,
                       total4, C, relu, ld_z, (g_bn_order >> 1) & 1);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

// statistics from slab sums written by the producer of x: slab (nslab, 2, C) = column sums of x and x^2 over consecutive row
// slabs (crb_group_affine_rows_stats_stack: 64 rows each). Block b of the usual statistics grid adds the slabs of its share in
// index order into the usual partial layout and the ticket finalize (or the finalize launch) follows as in the forward above.
__global__ __launch_bounds__(256) void bn_slab_partial_kernel(const float* __restrict__ slab, int64_t nslab, int C,
                                                              float* __restrict__ partial, BnFinal fin) {
  const int nblk = gridDim.x, blk = blockIdx.x;
  const int64_t per = (nslab + nblk - 1) / nblk;
  const int64_t s0 = (int64_t)blk * per, s1 = min(nslab, s0 + per);
  const int C2 = 2 * C;
  for (int i = threadIdx.x; i < C2; i += 256) {
    float a = 0.f;
    for (int64_t k = s0; k < s1; ++k) a += slab[k * C2 + i];
    if (fin.tickets) st_agent(partial + (int64_t)blk * C2 + i, a);
    else partial[(int64_t)blk * C2 + i] = a;
  }
  if (fin.tickets) bn_ticket_finalize(partial, nblk, C, blk, fin);
}

extern "C" int crb_bn_relu_forward_partials(const float* x, int64_t n, int C, const float* slab_sums, int64_t n_slabs,
                                            const float* gamma, const float* beta, float eps, int relu, float* z,
                                            int64_t z_row_stride, float* mean, float* var, float* invstd, float* running_mean,
                                            float* running_var, int64_t* num_batches_tracked, float momentum, void* workspace,
                                            int64_t workspace_bytes, int32_t* tickets, void* stream) {
  const int64_t ld_z = z_row_stride > 0 ? z_row_stride : C;
  if (ld_z < C || (ld_z & 3) || !slab_sums || n_slabs <= 0) return CRB_ERR_ARG;
  if (n <= 0 || C <= 0 || (C & 3) || C > 1024 || (256 % (C >> 2) && (C >> 2) < 256)) return CRB_ERR_ARG;
  if (workspace_bytes < crb_bn_workspace_bytes(n, C) - 256 || !workspace) return CRB_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = (int)(n_slabs < bn_blocks(n) ? n_slabs : bn_blocks(n));
  float* partial = (float*)workspace;
  hipLaunchKernelGGL(bn_slab_partial_kernel, dim3(nblk), dim3(256), 0, st, slab_sums, n_slabs, C, partial,
                     bn_final(tickets, workspace, n, C, 0, mean, var, invstd, running_mean, running_var, momentum, eps,
                              num_batches_tracked));
  if (!tickets)
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(crb_cdiv(C, 8)), dim3(256), 0, st, partial, nblk, C, n, eps, 0, mean, var,
                       invstd, running_mean, running_var, momentum, reinterpret_cast<long long*>(num_batches_tracked));
  const int64_t total4 = n * C / 4;
  if (z)                                       // z == NULL: statistics (and running statistics) only
    hipLaunchKernelGGL(bn_apply_kernel, dim3(crb_cdiv(total4, 256)), dim3(256), 0, st, x, mean, invstd, gamma, beta, z,
                       total4, C, relu, ld_z, (g_bn_order >> 1) & 1);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

// inference forward with given statistics (mean, invstd)
extern "C" int crb_bn_relu_apply(const float* x, int64_t n, int C, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, int relu, float* z, int64_t z_row_stride,
                                 void* stream) {
  if (n < 0 || C <= 0 || (C & 3)) return CRB_ERR_ARG;
  const int64_t ld_z = z_row_stride > 0 ? z_row_stride : C;
  if (ld_z < C || (ld_z & 3)) return CRB_ERR_ARG;
  if (n == 0) return CRB_OK;
  const int64_t total4 = n * C / 4;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(crb_cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream, x, mean, invstd,
                     gamma, beta, z, total4, C, relu, ld_z, 0);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_bn_relu_backward(const float* x, const float* dz, int64_t dz_row_stride, int64_t n, int C, const float* mean,
                                    const float* invstd, const float* gamma, const float* beta, int relu, float* dx,
                                    float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes, int32_t* tickets,
                                    void* stream) {
  if (n <= 0 || C <= 0 || (C & 3) || C > 1024 || (256 % (C >> 2) && (C >> 2) < 256)) return CRB_ERR_ARG;
  if (workspace_bytes < crb_bn_workspace_bytes(n, C) - 256 || !workspace) return CRB_ERR_WORKSPACE;
  const int64_t ld_dz = dz_row_stride > 0 ? dz_row_stride : C;
  if (ld_dz < C || (ld_dz & 3)) return CRB_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = bn_blocks(n);
  float* partial = (float*)workspace;
  hipLaunchKernelGGL(bn_partial_kernel<true>, dim3(nblk), dim3(256), 2 * 256 * 16, st, x, dz, mean, invstd, gamma, beta, n,
                     C, relu, bn_rows_per_block(n), ld_dz, partial, g_bn_order & 1,
                     bn_final(tickets, workspace, n, C, 1, dbeta, dgamma, nullptr, nullptr, nullptr, 0.f, 0.f));
  if (!tickets)
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(crb_cdiv(C, 8)), dim3(256), 0, st, partial, nblk, C, n, 0.f, 1, dbeta, dgamma,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, 0.f);
  const int64_t total4 = n * C / 4;
  if (dx)                      // dx == NULL: the reduced gradients only (the consumer applies the BatchNorm backward itself)
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(crb_cdiv(total4, 256)), dim3(256), 0, st, x, dz, mean, invstd, gamma, beta,
                       dbeta, dgamma, dx, total4, C, 1.0f / (float)n, relu, ld_dz, (g_bn_order >> 1) & 1);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

#ifdef CRB_MEASURE
// (measurement library only: the consumer side of crb_conv3x3_winograd2_bnbwd_nhwc, VERDICT r04 item 6a - measured, not adopted)
// backward with the reduction already done by the producer of dz: slab_sums (n_slabs, 2, C) = column sums of dz [z > 0] and
// dz [z > 0] xhat over disjoint row sets (crb_conv3x3_winograd2_bnbwd_nhwc) -> dbeta, dgamma by the ordered slab reduction + ticket
// finalize of crb_bn_relu_forward_partials, then the apply pass of crb_bn_relu_backward
extern "C" int crb_bn_relu_backward_partials(const float* x, const float* dz, int64_t dz_row_stride, int64_t n, int C,
                                             const float* slab_sums, int64_t n_slabs, const float* mean, const float* invstd,
                                             const float* gamma, const float* beta, int relu, float* dx, float* dgamma, float* dbeta,
                                             void* workspace, int64_t workspace_bytes, int32_t* tickets, void* stream) {
  if (n <= 0 || C <= 0 || (C & 3) || C > 1024 || (256 % (C >> 2) && (C >> 2) < 256) || !slab_sums || n_slabs <= 0) return CRB_ERR_ARG;
  if (workspace_bytes < crb_bn_workspace_bytes(n, C) - 256 || !workspace) return CRB_ERR_WORKSPACE;
  const int64_t ld_dz = dz_row_stride > 0 ? dz_row_stride : C;
  if (ld_dz < C || (ld_dz & 3)) return CRB_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = (int)(n_slabs < bn_blocks(n) ? n_slabs : bn_blocks(n));
  float* partial = (float*)workspace;
  hipLaunchKernelGGL(bn_slab_partial_kernel, dim3(nblk), dim3(256), 0, st, slab_sums, n_slabs, C, partial,
                     bn_final(tickets, workspace, n, C, 1, dbeta, dgamma, nullptr, nullptr, nullptr, 0.f, 0.f));
  if (!tickets)
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(crb_cdiv(C, 8)), dim3(256), 0, st, partial, nblk, C, n, 0.f, 1, dbeta, dgamma,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, 0.f);
  const int64_t total4 = n * C / 4;
  if (dx)
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(crb_cdiv(total4, 256)), dim3(256), 0, st, x, dz, mean, invstd, gamma, beta,
                       dbeta, dgamma, dx, total4, C, 1.0f / (float)n, relu, ld_dz, (g_bn_order >> 1) & 1);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}
#endif

// training BatchNorm + ReLU over all groups*ns rows, then max over each group of ns consecutive rows.
// zmax (groups, out_row_stride >= C) may be a column slice of a wider buffer; arg (groups, C) = first row of the group
// attaining the max. workspace: crb_bn_workspace_bytes(groups * ns, C).
extern "C" int crb_bn_relu_max_forward(const float* x, int64_t groups, int ns, int C, const float* gamma, const float* beta,
                                       float eps, float* zmax, int64_t out_row_stride, int32_t* arg, float* mean, float* var,
                                       float* invstd, float* running_mean, float* running_var,
                                       int64_t* num_batches_tracked, float momentum, void* workspace,
                                       int64_t workspace_bytes, int32_t* tickets, void* stream) {
  const int64_t n = groups * ns;
  const int64_t ld = out_row_stride > 0 ? out_row_stride : C;
  if (ld < C || (ld & 3) || ns <= 0) return CRB_ERR_ARG;
  if (n <= 0 || C <= 0 || (C & 3) || C > 1024 || (256 % (C >> 2) && (C >> 2) < 256)) return CRB_ERR_ARG;
  if (workspace_bytes < crb_bn_workspace_bytes(n, C) - 256 || !workspace) return CRB_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = bn_blocks(n);
  float* partial = (float*)workspace;
  hipLaunchKernelGGL(bn_partial_kernel<false>, dim3(nblk), dim3(256), 2 * 256 * 16, st, x, nullptr, nullptr, nullptr,
                     nullptr, nullptr, n, C, 1, bn_rows_per_block(n), (int64_t)C, partial, 0,
                     bn_final(tickets, workspace, n, C, 0, mean, var, invstd, running_mean, running_var, momentum, eps,
                              num_batches_tracked));
  if (!tickets)
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(crb_cdiv(C, 8)), dim3(256), 0, st, partial, nblk, C, n, eps, 0, mean, var,
                       invstd, running_mean, running_var, momentum, reinterpret_cast<long long*>(num_batches_tracked));
  const int glanes = 256 / (C >> 2) > 0 ? 256 / (C >> 2) : 1;
  hipLaunchKernelGGL(bn_relu_max_kernel, dim3(crb_cdiv(groups, glanes)), dim3(256), 0, st, x, mean, invstd, gamma, beta,
                     groups, ns, C, zmax, ld, arg);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

// backward of the above: gz (groups, gz_row_stride >= C) is the gradient w.r.t. zmax; dx (groups*ns, C) dense.
extern "C" int crb_bn_relu_max_backward(const float* x, const float* gz, int64_t gz_row_stride, const int32_t* arg,
                                        int64_t groups, int ns, int C, const float* mean, const float* invstd,
                                        const float* gamma, const float* beta, float* dx, float* dgamma, float* dbeta,
                                        void* workspace, int64_t workspace_bytes, int32_t* tickets, void* stream) {
  const int64_t n = groups * ns;
  const int64_t ld = gz_row_stride > 0 ? gz_row_stride : C;
  if (ld < C || (ld & 3) || ns <= 0) return CRB_ERR_ARG;
  if (n <= 0 || C <= 0 || (C & 3) || C > 1024 || (256 % (C >> 2) && (C >> 2) < 256)) return CRB_ERR_ARG;
  if (workspace_bytes < crb_bn_workspace_bytes(n, C) - 256 || !workspace) return CRB_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  if (n >= (1LL << 32) || 256 % (C >> 2)) return CRB_ERR_UNSUPPORTED;   // 32-bit row arithmetic in the apply pass
  const int gpb = bn_rows_per_block(groups);
  const int nblk = crb_cdiv(groups, gpb);                       // <= bn_blocks(n): fits the same workspace
  float* partial = (float*)workspace;
  hipLaunchKernelGGL(bn_max_partial_kernel, dim3(nblk), dim3(256), 2 * 256 * 16, st, x, gz, ld, arg, mean, invstd, gamma,
                     beta, groups, ns, C, gpb, partial,
                     bn_final(tickets, workspace, n, C, 1, dbeta, dgamma, nullptr, nullptr, nullptr, 0.f, 0.f));
  if (!tickets)
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(crb_cdiv(C, 8)), dim3(256), 0, st, partial, nblk, C, n, 0.f, 1, dbeta, dgamma,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, 0.f);
  const int64_t total4 = n * C / 4;
  hipLaunchKernelGGL(bn_max_bwd_apply_kernel, dim3(crb_cdiv(total4, 256)), dim3(256), 0, st, x, gz, ld, arg, mean, invstd,
                     gamma, beta, dbeta, dgamma, dx, total4, ns, C, 1.0f / (float)n);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_bn_relu_max_backward_sums(const float* x_sel, const float* gz, int64_t gz_row_stride, int64_t groups, int C,
                                             const float* mean, const float* invstd, const float* gamma, const float* beta,
                                             float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes,
                                             int32_t* tickets, void* stream) {
  const int64_t ld = gz_row_stride > 0 ? gz_row_stride : C;
  if (ld < C || (ld & 3) || !x_sel || !gz) return CRB_ERR_ARG;
  if (groups <= 0 || C <= 0 || (C & 3) || C > 1024 || 256 % (C >> 2)) return CRB_ERR_ARG;
  if (workspace_bytes < crb_bn_workspace_bytes(groups, C) - 256 || !workspace) return CRB_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int gpb = bn_rows_per_block(groups);
  const int nblk = crb_cdiv(groups, gpb);
  float* partial = (float*)workspace;
  hipLaunchKernelGGL(bn_max_partial_kernel, dim3(nblk), dim3(256), 2 * 256 * 16, st, x_sel, gz, ld, (const int*)nullptr, mean,
                     invstd, gamma, beta, groups, 1, C, gpb, partial,
                     bn_final(tickets, workspace, groups, C, 1, dbeta, dgamma, nullptr, nullptr, nullptr, 0.f, 0.f));
  if (!tickets)
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(crb_cdiv(C, 8)), dim3(256), 0, st, partial, nblk, C, groups, 0.f, 1, dbeta, dgamma,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, 0.f);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Per-frame statistics: the same launches once per row range of a frame (frame_row_offsets: HOST int64[n_frames+1]) — what
// n_frames separate train-mode passes of one frame each compute (batched CRB stage 2: every frame normalises with its own
// statistics, crb_sampling.py:174-212). The loop lives here so that a BatchNorm layer of a 16-frame pass costs one call from
// the host language instead of 16. The batch statistics themselves are scratch (first 3*C floats of the workspace);
// running statistics are updated once per frame, in frame order. workspace: crb_bn_frames_workspace_bytes(n_frames, max rows
// of a frame, C) = partials of every frame + 4 C statistics per frame.
extern "C" int64_t crb_bn_frames_workspace_bytes(int n_frames, int64_t max_rows_per_frame, int C) {
  const int64_t f = n_frames < 1 ? 1 : (n_frames > BN_MAXF ? BN_MAXF : n_frames);     // more frames run in chunks of BN_MAXF
  // bn_blocks(n) is not monotonic above 131072 rows (1024 blocks at 131072, 1017 at 131073): a batch that mixes frames just
  // under and just over needs the larger count for each -> size every frame for the maximum any row count <= max can take
  const int64_t blocks = max_rows_per_frame > 131072 ? 1024 : bn_blocks(max_rows_per_frame < 1 ? 1 : max_rows_per_frame);
  return f * (blocks * 2 * C * 4 + 256) + crb_align_up(f * 4 * C * (int64_t)sizeof(float), 256);
}

// statistics of up to BN_MAXF frames: partial sums, finalize, running update. stats (n_frames, 4, C) at the workspace head.
static int bn_frames_statistics(const float* x, int n_frames, const int64_t* off, int C, float eps, float* running_mean,
                                float* running_var, float momentum, void* workspace, int64_t workspace_bytes,
                                BnFrames& fr, float** stats_out, int64_t* max_rows_out, hipStream_t st) {
  if (n_frames <= 0 || n_frames > BN_MAXF || !off) return CRB_ERR_ARG;
  if (C <= 0 || (C & 3) || C > 1024 || (256 % (C >> 2) && (C >> 2) < 256)) return CRB_ERR_ARG;
  int slots = 0, max_blk = 0;
  int64_t max_rows = 0;
  for (int f = 0; f < n_frames; ++f) {
    const int64_t n = off[f + 1] - off[f];
    if (n < 0 || n == 1) return CRB_ERR_ARG;                 // a one-row batch has no variance (nn.BatchNorm raises too)
    fr.off[f] = off[f];
    fr.rpb[f] = n > 0 ? bn_rows_per_block(n) : 1;
    fr.nblk[f] = n > 0 ? bn_blocks(n) : 0;
    fr.pbase[f] = slots;
    slots += fr.nblk[f];
    max_blk = fr.nblk[f] > max_blk ? fr.nblk[f] : max_blk;
    max_rows = n > max_rows ? n : max_rows;
  }
  fr.off[n_frames] = off[n_frames];
  const int64_t head = crb_align_up((int64_t)n_frames * 4 * C * (int64_t)sizeof(float), 256);
  if (!workspace || workspace_bytes < head + (int64_t)slots * 2 * C * 4) return CRB_ERR_WORKSPACE;
  float* stats = (float*)workspace;
  float* partial = (float*)((char*)workspace + head);
  *stats_out = stats;
  *max_rows_out = max_rows;
  if (max_blk == 0) return CRB_OK;
  hipLaunchKernelGGL(bn_partial_frames_kernel, dim3(max_blk, n_frames), dim3(256), 2 * 256 * 16, st, x, C, fr, partial);
  hipLaunchKernelGGL(bn_finalize_frames_kernel, dim3(crb_cdiv(C, 8), n_frames), dim3(256), 0, st, partial, C, eps, fr, stats);
  if (running_mean && running_var)
    hipLaunchKernelGGL(bn_running_frames_kernel, dim3(crb_cdiv(C, 256)), dim3(256), 0, st, stats, C, n_frames, fr, momentum,
                       running_mean, running_var);
  return CRB_OK;
}

extern "C" int crb_bn_relu_forward_frames(const float* x, int n_frames, const int64_t* frame_row_offsets, int C,
                                          const float* gamma, const float* beta, float eps, int relu, float* z,
                                          int64_t z_row_stride, float* running_mean, float* running_var, float momentum,
                                          void* workspace, int64_t workspace_bytes, void* stream) {
  const int64_t ld_z = z_row_stride > 0 ? z_row_stride : C;
  if (ld_z < C || (ld_z & 3) || n_frames <= 0 || !frame_row_offsets) return CRB_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  for (int f0 = 0; f0 < n_frames; f0 += BN_MAXF) {           // offsets are absolute rows: x and z stay unshifted
    const int nf = n_frames - f0 < BN_MAXF ? n_frames - f0 : BN_MAXF;
    BnFrames fr;
    float* stats = nullptr;
    int64_t max_rows = 0;
    int rc = bn_frames_statistics(x, nf, frame_row_offsets + f0, C, eps, running_mean, running_var, momentum, workspace,
                                  workspace_bytes, fr, &stats, &max_rows, st);
    if (rc != CRB_OK) return rc;
    if (max_rows > 0)
      hipLaunchKernelGGL(bn_apply_frames_kernel, dim3(crb_cdiv(max_rows * C / 4, 256), nf), dim3(256), 0, st, x, stats, gamma,
                         beta, z, C, relu, ld_z, fr);
  }
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

// the StackSAModuleMSG tail (BatchNorm + ReLU + max over nsample) with per-frame statistics: every frame owns
// groups_per_frame consecutive query points
extern "C" int crb_bn_relu_max_forward_frames(const float* x, int n_frames, int64_t groups_per_frame, int ns, int C,
                                              const float* gamma, const float* beta, float eps, float* zmax,
                                              int64_t out_row_stride, int32_t* arg, float* running_mean, float* running_var,
                                              float momentum, void* workspace, int64_t workspace_bytes, void* stream) {
  if (n_frames <= 0 || groups_per_frame <= 0 || ns <= 0) return CRB_ERR_ARG;
  const int64_t ld = out_row_stride > 0 ? out_row_stride : C;
  if (ld < C || (ld & 3)) return CRB_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int glanes = 256 / (C >> 2) > 0 ? 256 / (C >> 2) : 1;
  for (int f0 = 0; f0 < n_frames; f0 += BN_MAXF) {
    const int nf = n_frames - f0 < BN_MAXF ? n_frames - f0 : BN_MAXF;
    const int64_t g0 = (int64_t)f0 * groups_per_frame;
    int64_t off[BN_MAXF + 1];
    for (int f = 0; f <= nf; ++f) off[f] = (int64_t)f * groups_per_frame * ns;            // relative to the chunk's rows
    BnFrames fr;
    float* stats = nullptr;
    int64_t max_rows = 0;
    int rc = bn_frames_statistics(x + g0 * ns * C, nf, off, C, eps, running_mean, running_var, momentum, workspace,
                                  workspace_bytes, fr, &stats, &max_rows, st);
    if (rc != CRB_OK) return rc;
    const int64_t groups = (int64_t)nf * groups_per_frame;
    hipLaunchKernelGGL(bn_relu_max_frames_kernel, dim3(crb_cdiv(groups, glanes)), dim3(256), 0, st, x + g0 * ns * C, stats,
                       gamma, beta, groups, groups_per_frame, ns, C, zmax + g0 * ld, ld, arg + g0 * C);
  }
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

#ifdef CRB_MEASURE
extern "C" int crb_bn_set_order(int bits) {
  g_bn_order = bits;
  return CRB_OK;
}
#endif

// (scale, shift) per channel of a training-mode BatchNorm whose statistics are known: z = scale * x + shift, interleaved (C, 2):
// what crb_conv3x3_winograd2_bnrelu_nhwc / crb_winograd2_wgrad_bnrelu apply inside their input transforms
namespace {
__global__ __launch_bounds__(256) void bn_affine_table_kernel(const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, int C,
                                                              float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float s = gamma[c] * invstd[c];
  out[2 * c] = s;
  out[2 * c + 1] = beta[c] - mean[c] * s;
}
}  // namespace

extern "C" int crb_bn_affine_table(const float* mean, const float* invstd, const float* gamma, const float* beta, int C, float* out,
                                   void* stream) {
  if (C <= 0 || !mean || !invstd || !gamma || !beta || !out) return CRB_ERR_ARG;
  hipLaunchKernelGGL(bn_affine_table_kernel, dim3(crb_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, mean, invstd, gamma, beta, C,
                     out);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}
