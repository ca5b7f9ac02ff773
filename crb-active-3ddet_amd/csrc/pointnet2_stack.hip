// PointNet++ "stack" ops for gfx950 (rows a16-a18 of SURVEY §8): farthest point sampling, ball query, grouping
// (+grad), three-NN, three-interpolate (+grad).
//
// Replaces pcdet/ops/pointnet2/pointnet2_stack/src/{sampling_gpu.cu:25-140, ball_query_gpu.cu:16-66,
// group_points_gpu.cu:15-102, interpolate_gpu.cu:16-172} (pybind surface pointnet2_api.cpp:12-31).
//
// Layout ("stacked"): xyz (N1+N2+..,3) f32 with *_batch_cnt (B) i32; indices are int32.
//
// MI355X design notes
//  * ball query: ONE WAVE per query centre. The 64 lanes test 64 consecutive candidate points per step (coalesced
//    12-B rows), a ballot + popcount prefix places the hits in index order, and the wave stops as soon as nsample hits
//    exist — the reference's per-thread serial scan order (ball_query_gpu.cu:47-64) is reproduced exactly, 64 points
//    per step.
//  * FPS: one 1024-thread workgroup per frame with every point of the frame held in registers for the whole run;
//    each round is a register-only distance update + a wave arg-max by __shfl_xor on a packed 64-bit key + ONE LDS
//    exchange between the 16 waves (2 barriers per round instead of the reference's 11). The packed key encodes the
//    reference's tie rule exactly (first max per strided thread, then the lower-position operand of its LDS tree).
//  * grouping: one workgroup per centre tile, rows read coalesced along C, the (C,nsample) slab transposed through LDS
//    so the (M,C,nsample) output is written contiguously; the gradient scatters with channel-contiguous atomics.
#include <hipcub/hipcub.hpp>
#include "crb_common.h"
#include "../../include/crb_hip.h"

namespace {

// which batch element does stacked row `i` belong to; also returns the start row of that element in `other_cnt`
__device__ __forceinline__ int locate_batch(const int* __restrict__ cnt, int B, int i, const int* __restrict__ other_cnt,
                                            int* other_start) {
  int b = 0, acc = cnt[0], os = 0;
  for (int k = 1; k < B; ++k) {
    if (i < acc) break;
    acc += cnt[k];
    os += other_cnt[k - 1];
    b = k;
  }
  *other_start = os;
  return b;
}

// ------------------------------------------------------------------------------------------------ ball query
__global__ __launch_bounds__(256) void ball_query_kernel(int B, int M, float radius, int nsample,
                                                         const float* __restrict__ new_xyz,
                                                         const int* __restrict__ new_cnt, const float* __restrict__ xyz,
                                                         const int* __restrict__ xyz_cnt, int* __restrict__ idx) {
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (q >= M) return;
  int start;
  const int b = locate_batch(new_cnt, B, q, xyz_cnt, &start);
  const int n = xyz_cnt[b];
  const float* p = xyz + (int64_t)start * 3;
  const float r2 = radius * radius;
  const float qx = new_xyz[(int64_t)q * 3 + 0], qy = new_xyz[(int64_t)q * 3 + 1], qz = new_xyz[(int64_t)q * 3 + 2];
  int* out = idx + (int64_t)q * nsample;
  int cnt = 0;
  int first = -1;
  for (int k0 = 0; k0 < n && cnt < nsample; k0 += 64) {
    const int k = k0 + lane;
    bool hit = false;
    if (k < n) {
      float x = p[k * 3 + 0], y = p[k * 3 + 1], z = p[k * 3 + 2];
      float d2 = (qx - x) * (qx - x) + (qy - y) * (qy - y) + (qz - z) * (qz - z);
      hit = d2 < r2;
    }
    const unsigned long long m = __ballot(hit);
    if (m == 0ULL) continue;
    if (first < 0) first = k0 + (__ffsll((long long)m) - 1);
    const int pos = cnt + __popcll(m & ((1ULL << lane) - 1ULL));
    if (hit && pos < nsample) out[pos] = k;
    cnt += __popcll(m);
  }
  if (cnt > nsample) cnt = nsample;
  // pad with the first hit (the reference pre-fills every slot with it), or flag an empty ball with -1 in slot 0
  for (int l = cnt + lane; l < nsample; l += 64) out[l] = (first >= 0) ? first : 0;
  if (first < 0 && lane == 0) out[0] = -1;
}

// Two radii in ONE scan of the frame's points (StackSAModuleMSG always queries the same centres with two radii): the
// distance of a candidate is computed once and tested against both; the scan stops when both lists are full. Writes the
// index lists already in the form the grouping kernels consume (empty ball -> all zeros) plus the empty flags, so the three
// torch launches of the reference's fix-up (pointnet2_utils.py:36-38) disappear as well.
__global__ __launch_bounds__(256) void ball_query2_kernel(int B, int M, float ra, int nsa, float rb, int nsb,
                                                          const float* __restrict__ new_xyz, const int* __restrict__ new_cnt,
                                                          const float* __restrict__ xyz, const int* __restrict__ xyz_cnt,
                                                          int* __restrict__ idx_a, int* __restrict__ idx_b,
                                                          unsigned char* __restrict__ empty_a,
                                                          unsigned char* __restrict__ empty_b) {
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (q >= M) return;
  int start;
  const int b = locate_batch(new_cnt, B, q, xyz_cnt, &start);
  const int n = xyz_cnt[b];
  const float* p = xyz + (int64_t)start * 3;
  const float ra2 = ra * ra, rb2 = rb * rb;
  const float qx = new_xyz[(int64_t)q * 3 + 0], qy = new_xyz[(int64_t)q * 3 + 1], qz = new_xyz[(int64_t)q * 3 + 2];
  int* oa = idx_a + (int64_t)q * nsa;
  int* ob = idx_b + (int64_t)q * nsb;
  int ca = 0, cb = 0, fa = -1, fb = -1;
  for (int k0 = 0; k0 < n && (ca < nsa || cb < nsb); k0 += 64) {
    const int k = k0 + lane;
    bool ha = false, hb = false;
    if (k < n) {
      float x = p[k * 3 + 0], y = p[k * 3 + 1], z = p[k * 3 + 2];
      float d2 = (qx - x) * (qx - x) + (qy - y) * (qy - y) + (qz - z) * (qz - z);
      ha = d2 < ra2;
      hb = d2 < rb2;
    }
    const unsigned long long ma = __ballot(ha), mb = __ballot(hb);
    const unsigned long long below = (1ULL << lane) - 1ULL;
    if (ma != 0ULL && ca < nsa) {
      if (fa < 0) fa = k0 + (__ffsll((long long)ma) - 1);
      const int pos = ca + __popcll(ma & below);
      if (ha && pos < nsa) oa[pos] = k;
      ca += __popcll(ma);
    }
    if (mb != 0ULL && cb < nsb) {
      if (fb < 0) fb = k0 + (__ffsll((long long)mb) - 1);
      const int pos = cb + __popcll(mb & below);
      if (hb && pos < nsb) ob[pos] = k;
      cb += __popcll(mb);
    }
  }
  ca = ca > nsa ? nsa : ca;
  cb = cb > nsb ? nsb : cb;
  for (int l = ca + lane; l < nsa; l += 64) oa[l] = (fa >= 0) ? fa : 0;
  for (int l = cb + lane; l < nsb; l += 64) ob[l] = (fb >= 0) ? fb : 0;
  if (lane == 0) {
    empty_a[q] = fa < 0;
    empty_b[q] = fb < 0;
  }
}

// Same contract, Q consecutive queries per wave. With one query per wave every wave streams its frame's points through the
// caches on its own (VSA: 32768 queries x 240 KB = 7.9 GB of L2 reads per call, 2.4 distance tests per clock and CU); here
// the 64 candidates of a step are loaded once and tested against the wave's Q queries (wave-uniform coordinates and
// counters), so the kernel runs at the VALU rate of the distance test. Queries of one wave must lie in one frame; a wave
// that straddles a frame boundary walks its queries one by one.
template <int Q>
__global__ __launch_bounds__(256) void ball_query2_multi_kernel(int B, int M, float ra, int nsa, float rb, int nsb,
                                                                const float* __restrict__ new_xyz,
                                                                const int* __restrict__ new_cnt,
                                                                const float* __restrict__ xyz, const int* __restrict__ xyz_cnt,
                                                                int* __restrict__ idx_a, int* __restrict__ idx_b,
                                                                unsigned char* __restrict__ empty_a,
                                                                unsigned char* __restrict__ empty_b) {
  // In the pipelined scoring pass these waves share the sampling kernel's 16 CUs with the NEXT batch's farthest-point sampling (off
  // the critical path, a batch ahead). All workgroups of a call are resident at once, so the ones on those CUs set the kernel's
  // time: beside fps2_kernel (few stalls) a call took 646 us instead of 244. Issue priority over the default-priority sampler:
  __builtin_amdgcn_s_setprio(3);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int q0 = (blockIdx.x * 4 + wave) * Q;
  if (q0 >= M) return;
  const int nq = min(Q, M - q0);
  const float ra2 = ra * ra, rb2 = rb * rb;
  const unsigned long long below = (1ULL << lane) - 1ULL;
  int start0, start1;
  const int b0 = locate_batch(new_cnt, B, q0, xyz_cnt, &start0);
  const int b1 = locate_batch(new_cnt, B, q0 + nq - 1, xyz_cnt, &start1);
  if (b0 != b1) {                                            // frame boundary inside the group: one query at a time
    for (int i = 0; i < nq; ++i) {
      const int q = q0 + i;
      int start;
      const int b = locate_batch(new_cnt, B, q, xyz_cnt, &start);
      const int n = xyz_cnt[b];
      const float* p = xyz + (int64_t)start * 3;
      const float qx = new_xyz[(int64_t)q * 3 + 0], qy = new_xyz[(int64_t)q * 3 + 1], qz = new_xyz[(int64_t)q * 3 + 2];
      int* oa = idx_a + (int64_t)q * nsa;
      int* ob = idx_b + (int64_t)q * nsb;
      int ca = 0, cb = 0, fa = -1, fb = -1;
      for (int k0 = 0; k0 < n && (ca < nsa || cb < nsb); k0 += 64) {
        const int k = k0 + lane;
        bool ha = false, hb = false;
        if (k < n) {
          float x = p[k * 3 + 0], y = p[k * 3 + 1], z = p[k * 3 + 2];
          float d2 = (qx - x) * (qx - x) + (qy - y) * (qy - y) + (qz - z) * (qz - z);
          ha = d2 < ra2;
          hb = d2 < rb2;
        }
        const unsigned long long ma = __ballot(ha), mb = __ballot(hb);
        if (ma != 0ULL && ca < nsa) {
          if (fa < 0) fa = k0 + (__ffsll((long long)ma) - 1);
          const int pos = ca + __popcll(ma & below);
          if (ha && pos < nsa) oa[pos] = k;
          ca += __popcll(ma);
        }
        if (mb != 0ULL && cb < nsb) {
          if (fb < 0) fb = k0 + (__ffsll((long long)mb) - 1);
          const int pos = cb + __popcll(mb & below);
          if (hb && pos < nsb) ob[pos] = k;
          cb += __popcll(mb);
        }
      }
      ca = ca > nsa ? nsa : ca;
      cb = cb > nsb ? nsb : cb;
      for (int l = ca + lane; l < nsa; l += 64) oa[l] = (fa >= 0) ? fa : 0;
      for (int l = cb + lane; l < nsb; l += 64) ob[l] = (fb >= 0) ? fb : 0;
      if (lane == 0) { empty_a[q] = fa < 0; empty_b[q] = fb < 0; }
    }
    return;
  }
  const int n = xyz_cnt[b0];
  const float* p = xyz + (int64_t)start0 * 3;
  // (two queries per v_pk_add/mul_f32 instruction: measured slower, 295 vs 247 us per VSA call)
  float qx[Q], qy[Q], qz[Q];
  int ca[Q], cb[Q], fa[Q], fb[Q];
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    const int q = q0 + (i < nq ? i : 0);
    qx[i] = new_xyz[(int64_t)q * 3 + 0]; qy[i] = new_xyz[(int64_t)q * 3 + 1]; qz[i] = new_xyz[(int64_t)q * 3 + 2];
    ca[i] = i < nq ? 0 : nsa;                                // slots past the end count as finished
    cb[i] = i < nq ? 0 : nsb;
    fa[i] = fb[i] = -1;
  }
  // The common step has no hit for any of the Q queries: all Q distance tests run straight-line and ONE branch on the OR of
  // their ballots skips the bookkeeping (a branch per query cost as much as the arithmetic). A finished query (or an unused
  // slot) gets an infinite x coordinate: its d2 is +inf and never passes a radius test.
#pragma unroll
  for (int i = 0; i < Q; ++i)
    if (i >= nq) qx[i] = __builtin_inff();
  int open = nq;                                             // queries whose two lists are not both full
  for (int k0 = 0; k0 < n && open > 0; k0 += 64) {
    const int k = k0 + lane;
    const bool in = k < n;
    float x = 0.f, y = 0.f, z = 0.f;
    if (in) { x = p[k * 3 + 0]; y = p[k * 3 + 1]; z = p[k * 3 + 2]; }
    float d2[Q];
    unsigned long long mb[Q], any = 0ULL;
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      d2[i] = (qx[i] - x) * (qx[i] - x) + (qy[i] - y) * (qy[i] - y) + (qz[i] - z) * (qz[i] - z);
      mb[i] = __ballot(in && d2[i] < rb2);
      any |= mb[i];
    }
    if (any == 0ULL) continue;                               // launched with ra <= rb only: no hit of b, no hit of a
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      if (mb[i] == 0ULL) continue;
      const bool hb = in && d2[i] < rb2, ha = in && d2[i] < ra2;
      const unsigned long long ma = __ballot(ha);
      if (ma != 0ULL && ca[i] < nsa) {
        if (fa[i] < 0) fa[i] = k0 + (__ffsll((long long)ma) - 1);
        const int pos = ca[i] + __popcll(ma & below);
        if (ha && pos < nsa) idx_a[(int64_t)(q0 + i) * nsa + pos] = k;
        ca[i] += __popcll(ma);
      }
      if (cb[i] < nsb) {
        if (fb[i] < 0) fb[i] = k0 + (__ffsll((long long)mb[i]) - 1);
        const int pos = cb[i] + __popcll(mb[i] & below);
        if (hb && pos < nsb) idx_b[(int64_t)(q0 + i) * nsb + pos] = k;
        cb[i] += __popcll(mb[i]);
      }
      if (ca[i] >= nsa && cb[i] >= nsb) { --open; qx[i] = __builtin_inff(); }
    }
  }
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    if (i >= nq) break;
    const int q = q0 + i;
    const int na = ca[i] > nsa ? nsa : ca[i], nb = cb[i] > nsb ? nsb : cb[i];
    for (int l = na + lane; l < nsa; l += 64) idx_a[(int64_t)q * nsa + l] = (fa[i] >= 0) ? fa[i] : 0;
    for (int l = nb + lane; l < nsb; l += 64) idx_b[(int64_t)q * nsb + l] = (fb[i] >= 0) ? fb[i] : 0;
    if (lane == 0) { empty_a[q] = fa[i] < 0; empty_b[q] = fb[i] < 0; }
  }
}

// Same contract again for queries that come in spatially compact GROUPS of `group` consecutive rows inside one frame (the
// 216 grid points of one RoI, pvrcnn_head.py:97-132): one workgroup per group first keeps, in index order, the frame points
// within (group radius + larger ball radius) of the group's centroid in LDS — a point outside that sphere cannot be in any
// of the group's balls (triangle inequality, 0.1 % slack for rounding) — and every query then scans only those. At the
// RoI-grid shape 2048 keypoints shrink to a few dozen candidates per RoI. Index-exact with ball_query2_kernel: the candidate
// list preserves the scan order. A group whose candidates do not fit GQ_CAP falls back to scanning the frame.
constexpr int GQ_CAP = 2048;

__global__ __launch_bounds__(256) void ball_query2_grouped_kernel(int B, int M, int group, float ra, int nsa, float rb, int nsb,
                                                                  const float* __restrict__ new_xyz,
                                                                  const int* __restrict__ new_cnt,
                                                                  const float* __restrict__ xyz, const int* __restrict__ xyz_cnt,
                                                                  int* __restrict__ idx_a, int* __restrict__ idx_b,
                                                                  unsigned char* __restrict__ empty_a,
                                                                  unsigned char* __restrict__ empty_b) {
  __shared__ float cpos[GQ_CAP][3];
  __shared__ int cidx[GQ_CAP];
  __shared__ float red[4][4];
  __shared__ int wcount[4];
  __builtin_amdgcn_s_setprio(3);                             // (see ball_query2_multi_kernel)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int q0 = blockIdx.x * group;
  int start, start_last;
  const int b = locate_batch(new_cnt, B, q0, xyz_cnt, &start);
  const bool straddle = locate_batch(new_cnt, B, q0 + group - 1, xyz_cnt, &start_last) != b;   // caller broke its promise
  int n = xyz_cnt[b];
  const float* p = xyz + (int64_t)start * 3;
  const unsigned long long below = (1ULL << lane) - 1ULL;

  // centroid and radius of the group (group <= 1024: strided over the threads)
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int t = threadIdx.x; t < group; t += 256) {
    sx += new_xyz[(int64_t)(q0 + t) * 3 + 0]; sy += new_xyz[(int64_t)(q0 + t) * 3 + 1]; sz += new_xyz[(int64_t)(q0 + t) * 3 + 2];
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { sx += __shfl_xor(sx, d); sy += __shfl_xor(sy, d); sz += __shfl_xor(sz, d); }
  if (lane == 0) { red[wave][0] = sx; red[wave][1] = sy; red[wave][2] = sz; }
  __syncthreads();
  const float cx = (red[0][0] + red[1][0] + red[2][0] + red[3][0]) / (float)group;
  const float cy = (red[0][1] + red[1][1] + red[2][1] + red[3][1]) / (float)group;
  const float cz = (red[0][2] + red[1][2] + red[2][2] + red[3][2]) / (float)group;
  float r2 = 0.f;
  for (int t = threadIdx.x; t < group; t += 256) {
    const float dx = new_xyz[(int64_t)(q0 + t) * 3 + 0] - cx, dy = new_xyz[(int64_t)(q0 + t) * 3 + 1] - cy,
                dz = new_xyz[(int64_t)(q0 + t) * 3 + 2] - cz;
    r2 = fmaxf(r2, dx * dx + dy * dy + dz * dz);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, d));
  if (lane == 0) red[wave][3] = r2;
  __syncthreads();
  const float R = sqrtf(fmaxf(fmaxf(red[0][3], red[1][3]), fmaxf(red[2][3], red[3][3])));
  const float rmax = fmaxf(ra, rb);
  const float keep2 = (R + rmax) * (R + rmax) * 1.002f + 1e-6f;

  // order-preserving compaction of the frame points inside the sphere
  int total = 0;
  bool overflow = false;
  for (int k0 = 0; k0 < n; k0 += 256) {
    const int k = k0 + threadIdx.x;
    float x = 0.f, y = 0.f, z = 0.f;
    bool keep = false;
    if (k < n) {
      x = p[k * 3 + 0]; y = p[k * 3 + 1]; z = p[k * 3 + 2];
      keep = (x - cx) * (x - cx) + (y - cy) * (y - cy) + (z - cz) * (z - cz) <= keep2;
    }
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wcount[wave] = __popcll(m);
    __syncthreads();
    int off = total;
    for (int w = 0; w < wave; ++w) off += wcount[w];
    const int step_total = wcount[0] + wcount[1] + wcount[2] + wcount[3];
    const int pos = off + __popcll(m & below);
    if (keep && pos < GQ_CAP) { cidx[pos] = k; cpos[pos][0] = x; cpos[pos][1] = y; cpos[pos][2] = z; }
    total += step_total;
    __syncthreads();
  }
  overflow = total > GQ_CAP || straddle;          // a group across two frames: every query scans its own frame
  int nc = overflow ? n : total;
  const float ra2 = ra * ra, rb2 = rb * rb;
  for (int t = wave; t < group; t += 4) {
    const int q = q0 + t;
    if (straddle) {
      int st;
      const int bq = locate_batch(new_cnt, B, q, xyz_cnt, &st);
      n = xyz_cnt[bq];
      nc = n;
      p = xyz + (int64_t)st * 3;
    }
    const float qx = new_xyz[(int64_t)q * 3 + 0], qy = new_xyz[(int64_t)q * 3 + 1], qz = new_xyz[(int64_t)q * 3 + 2];
    int* oa = idx_a + (int64_t)q * nsa;
    int* ob = idx_b + (int64_t)q * nsb;
    int ca = 0, cb = 0, fa = -1, fb = -1;
    for (int k0 = 0; k0 < nc && (ca < nsa || cb < nsb); k0 += 64) {
      const int k = k0 + lane;
      bool ha = false, hb = false;
      int gi = 0;
      if (k < nc) {
        float x, y, z;
        if (overflow) { gi = k; x = p[k * 3 + 0]; y = p[k * 3 + 1]; z = p[k * 3 + 2]; }
        else { gi = cidx[k]; x = cpos[k][0]; y = cpos[k][1]; z = cpos[k][2]; }
        const float d2 = (qx - x) * (qx - x) + (qy - y) * (qy - y) + (qz - z) * (qz - z);
        ha = d2 < ra2;
        hb = d2 < rb2;
      }
      const unsigned long long ma = __ballot(ha), mb = __ballot(hb);
      if (ma != 0ULL && ca < nsa) {
        if (fa < 0) fa = __shfl(gi, __ffsll((long long)ma) - 1);
        const int pos = ca + __popcll(ma & below);
        if (ha && pos < nsa) oa[pos] = gi;
        ca += __popcll(ma);
      }
      if (mb != 0ULL && cb < nsb) {
        if (fb < 0) fb = __shfl(gi, __ffsll((long long)mb) - 1);
        const int pos = cb + __popcll(mb & below);
        if (hb && pos < nsb) ob[pos] = gi;
        cb += __popcll(mb);
      }
    }
    ca = ca > nsa ? nsa : ca;
    cb = cb > nsb ? nsb : cb;
    for (int l = ca + lane; l < nsa; l += 64) oa[l] = (fa >= 0) ? fa : 0;
    for (int l = cb + lane; l < nsb; l += 64) ob[l] = (fb >= 0) ? fb : 0;
    if (lane == 0) { empty_a[q] = fa < 0; empty_b[q] = fb < 0; }
  }
}

// ------------------------------------------------------------------------------------------------ grouping
// out (M, C, ns) ; one workgroup per centre; LDS slab (ns, C+1)
__global__ __launch_bounds__(256) void group_points_kernel(int B, int M, int C, int ns, const float* __restrict__ feat,
                                                           const int* __restrict__ feat_cnt, const int* __restrict__ idx,
                                                           const int* __restrict__ idx_cnt, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* slab = reinterpret_cast<float*>(smem);
  const int m = blockIdx.x;
  int start;
  locate_batch(idx_cnt, B, m, feat_cnt, &start);
  const int CP = C + 1;
  for (int t = threadIdx.x; t < ns * C; t += 256) {
    int s = t / C, c = t - s * C;
    int row = start + idx[(int64_t)m * ns + s];
    slab[s * CP + c] = feat[(int64_t)row * C + c];
  }
  __syncthreads();
  float* o = out + (int64_t)m * C * ns;
  for (int t = threadIdx.x; t < ns * C; t += 256) {
    int c = t / ns, s = t - c * ns;
    o[t] = slab[s * CP + c];
  }
}

__global__ __launch_bounds__(256) void group_points_grad_kernel(int B, int M, int C, int ns,
                                                                const float* __restrict__ grad_out,
                                                                const int* __restrict__ idx,
                                                                const int* __restrict__ idx_cnt,
                                                                const int* __restrict__ feat_cnt,
                                                                float* __restrict__ grad_feat) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* slab = reinterpret_cast<float*>(smem);
  const int m = blockIdx.x;
  int start;
  locate_batch(idx_cnt, B, m, feat_cnt, &start);
  const int CP = C + 1;
  const float* g = grad_out + (int64_t)m * C * ns;
  for (int t = threadIdx.x; t < ns * C; t += 256) {
    int c = t / ns, s = t - c * ns;
    slab[s * CP + c] = g[t];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < ns * C; t += 256) {
    int s = t / C, c = t - s * C;
    int row = start + idx[(int64_t)m * ns + s];
    atomicAdd(&grad_feat[(int64_t)row * C + c], slab[s * CP + c]);
  }
}

// ------------------------------------------------------------------------------------------------ fused QueryAndGroup
// out (3+C, M, ns) channel-major = the (1, 3+C, M, ns) NCHW tensor the shared MLP consumes:
//   c < 3 : xyz[nbr] - new_xyz   ;   c >= 3 : features[nbr][c-3]   ; all zero for an empty ball
// One launch replaces group(xyz), the centre subtraction, two mask multiplies, group(features), the concat and the
// (M,C,ns)->(1,C,M,ns) permute copy of QueryAndGroup + StackSAModuleMSG.forward (pointnet2_utils.py:129-155,
// pointnet2_modules.py:90-97) — seven passes over a GB-sized tensor at the RoI-grid pooling shapes.
// A 256-thread workgroup owns 64 consecutive (query, sample) pairs: rows are read coalesced along the channel axis into an
// LDS slab and written back 64-wide along the pair axis.
__global__ __launch_bounds__(256) void query_group_kernel(int B, int64_t MP /* M*ns */, int C, int ns,
                                                          const float* __restrict__ xyz, const int* __restrict__ xyz_cnt,
                                                          const float* __restrict__ feat, const float* __restrict__ new_xyz,
                                                          const int* __restrict__ new_cnt, const int* __restrict__ idx,
                                                          const unsigned char* __restrict__ empty, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* slab = reinterpret_cast<float*>(smem);            // 64 x (CT+1)
  __shared__ int srow[64];
  __shared__ int sm[64];
  const int CT = C + 3, CP = CT + 1;
  const int64_t p0 = (int64_t)blockIdx.x * 64;
  if (threadIdx.x < 64) {
    const int64_t p = p0 + threadIdx.x;
    int row = -1, m = 0;
    if (p < MP) {
      m = (int)(p / ns);
      if (!empty[m]) {
        int start;
        locate_batch(new_cnt, B, m, xyz_cnt, &start);
        row = start + idx[p];
      }
    }
    srow[threadIdx.x] = row;
    sm[threadIdx.x] = m;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * CT; e += 256) {
    const int pl = e / CT, c = e - pl * CT;
    const int row = srow[pl];
    float v = 0.f;
    if (row >= 0) v = (c < 3) ? xyz[(int64_t)row * 3 + c] - new_xyz[(int64_t)sm[pl] * 3 + c] : feat[(int64_t)row * C + c - 3];
    slab[pl * CP + c] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * CT; e += 256) {
    const int c = e >> 6, pl = e & 63;
    if (p0 + pl < MP) out[(int64_t)c * MP + p0 + pl] = slab[pl * CP + c];
  }
}

// grad_feat[row][c-3] += grad_out[c][p] for non-empty balls (c >= 3)
__global__ __launch_bounds__(256) void query_group_grad_kernel(int B, int64_t MP, int C, int ns,
                                                               const int* __restrict__ xyz_cnt,
                                                               const int* __restrict__ new_cnt, const int* __restrict__ idx,
                                                               const unsigned char* __restrict__ empty,
                                                               const float* __restrict__ grad_out,
                                                               float* __restrict__ grad_feat) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* slab = reinterpret_cast<float*>(smem);            // 64 x (C+1)
  __shared__ int srow[64];
  const int CP = C + 1;
  const int64_t p0 = (int64_t)blockIdx.x * 64;
  if (threadIdx.x < 64) {
    const int64_t p = p0 + threadIdx.x;
    int row = -1;
    if (p < MP) {
      const int m = (int)(p / ns);
      if (!empty[m]) {
        int start;
        locate_batch(new_cnt, B, m, xyz_cnt, &start);
        row = start + idx[p];
      }
    }
    srow[threadIdx.x] = row;
  }
  for (int e = threadIdx.x; e < 64 * C; e += 256) {
    const int c = e >> 6, pl = e & 63;
    slab[pl * CP + c] = (p0 + pl < MP) ? grad_out[(int64_t)(c + 3) * MP + p0 + pl] : 0.f;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * C; e += 256) {
    const int pl = e / C, c = e - pl * C;
    const int row = srow[pl];
    if (row >= 0) atomicAdd(&grad_feat[(int64_t)row * C + c], slab[pl * CP + c]);
  }
}

// Row-major variant for the training path: out (M*ns, 3+C). The shared MLP then runs as plain GEMMs on (pairs, channels)
// matrices followed by the fused BatchNorm+ReLU row kernels (bn_relu.hip) — MIOpen's BatchNorm2d on the (1,C,M,ns) view and
// the NCHW<->NHWC transposes around its 1x1 convs were 60 ms of a 247 ms PV-RCNN step.
__global__ __launch_bounds__(256) void query_group_rows_kernel(int B, int64_t MP, int C, int ns,
                                                               const float* __restrict__ xyz, const int* __restrict__ xyz_cnt,
                                                               const float* __restrict__ feat, const float* __restrict__ new_xyz,
                                                               const int* __restrict__ new_cnt, const int* __restrict__ idx,
                                                               const unsigned char* __restrict__ empty, float* __restrict__ out) {
  __shared__ int srow[64];
  __shared__ int sm[64];
  const int CT = C + 3;
  const int64_t p0 = (int64_t)blockIdx.x * 64;
  if (threadIdx.x < 64) {
    const int64_t p = p0 + threadIdx.x;
    int row = -1, m = 0;
    if (p < MP) {
      m = (int)(p / ns);
      if (!empty[m]) {
        int start;
        locate_batch(new_cnt, B, m, xyz_cnt, &start);
        row = start + idx[p];
      }
    }
    srow[threadIdx.x] = row;
    sm[threadIdx.x] = m;
  }
  __syncthreads();
  const int npl = (int)min((int64_t)64, MP - p0);
  float* dst = out + p0 * CT;
  for (int e = threadIdx.x; e < npl * CT; e += 256) {
    const int pl = e / CT, c = e - pl * CT;
    const int row = srow[pl];
    float v = 0.f;
    if (row >= 0) v = (c < 3) ? xyz[(int64_t)row * 3 + c] - new_xyz[(int64_t)sm[pl] * 3 + c] : feat[(int64_t)row * C + c - 3];
    dst[e] = v;
  }
}

// grad_feat[row][c-3] += grad_out[p][c] (c >= 3). The 64 pairs of a workgroup are 64/ns whole queries when ns divides 64, and
// a ball query pads its result by repeating the first hit, so most pairs of a slab share their source row with an earlier
// pair: those are summed inside LDS (in pair order) and only the first pair of each distinct row issues the atomics.
__global__ __launch_bounds__(256) void query_group_rows_grad_kernel(int B, int64_t MP, int C, int ns,
                                                                    const int* __restrict__ xyz_cnt,
                                                                    const int* __restrict__ new_cnt, const int* __restrict__ idx,
                                                                    const unsigned char* __restrict__ empty,
                                                                    const float* __restrict__ grad_out,
                                                                    float* __restrict__ grad_feat) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* slab = reinterpret_cast<float*>(smem);            // 64 x C
  __shared__ int srow[64];
  __shared__ int first[64];
  const int CT = C + 3;
  const int64_t p0 = (int64_t)blockIdx.x * 64;
  if (threadIdx.x < 64) {
    const int64_t p = p0 + threadIdx.x;
    int row = -1;
    if (p < MP) {
      const int m = (int)(p / ns);
      if (!empty[m]) {
        int start;
        locate_batch(new_cnt, B, m, xyz_cnt, &start);
        row = start + idx[p];
      }
    }
    srow[threadIdx.x] = row;
  }
  const int npl = (int)min((int64_t)64, MP - p0);
  for (int e = threadIdx.x; e < 64 * C; e += 256) {
    const int pl = e / C, c = e - pl * C;
    slab[e] = pl < npl ? grad_out[(p0 + pl) * CT + 3 + c] : 0.f;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int row = srow[threadIdx.x];
    int f = threadIdx.x;
    if (row >= 0)
      for (int q = 0; q < (int)threadIdx.x; ++q)
        if (srow[q] == row) { f = q; break; }
    first[threadIdx.x] = f;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {             // one thread per channel walks the pairs in order
    for (int pl = 0; pl < 64; ++pl) {
      const int f = first[pl];
      if (f != pl) slab[f * C + c] += slab[pl * C + c];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * C; e += 256) {
    const int pl = e / C, c = e - pl * C;
    const int row = srow[pl];
    if (row >= 0 && first[pl] == pl) atomicAdd(&grad_feat[(int64_t)row * C + c], slab[e]);
  }
}

// First shared-MLP layer of the training path without the grouped matrix. A bias-free 1x1 conv on [xyz_j - c_i ; f_j] is
//   y[p] = W1x (xyz_j - c_i) + (F W1f^T)[j]:  P = F W1f^T is one small GEMM over the N source points (caller), and this kernel
// gathers P rows and adds the 3-term offset part: the (pairs, 3+C) matrix (3.7 GB at the RoI-grid shape) and the
// pairs x (3+C) x H GEMM on it are never formed. rel (pairs,3) keeps xyz_j - c_i for the weight gradient.
typedef float gf4 __attribute__((ext_vector_type(4)));

// HT > 0: compile-time H (16/32/64/128), one float4 of an output row per thread; HT = 0: any H, scalar lanes
template <int HT>
__global__ __launch_bounds__(256) void group_affine_rows_kernel(int B, int64_t MP, int Hrt, int ns,
                                                                const float* __restrict__ xyz, const int* __restrict__ xyz_cnt,
                                                                const float* __restrict__ P, const float* __restrict__ new_xyz,
                                                                const int* __restrict__ new_cnt, const int* __restrict__ idx,
                                                                const unsigned char* __restrict__ empty,
                                                                const float* __restrict__ W1x /* (3,H) */,
                                                                float* __restrict__ out, float* __restrict__ rel,
                                                                float* __restrict__ stat = nullptr) {
  // stat (blocks, 2, H), HT > 0 only: this slab's column sums of out and out^2 — the statistics pass of the BatchNorm that
  // follows then reads 2 H floats per 64 rows instead of the rows themselves (crb_bn_relu_forward_partials)
  __shared__ int srow[64];
  __shared__ float sd[64][3];
  const int64_t p0 = (int64_t)blockIdx.x * 64;
  if (threadIdx.x < 64) {
    const int64_t p = p0 + threadIdx.x;
    int row = -1;
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (p < MP) {
      const int m = (int)(p / ns);
      if (!empty[m]) {
        int start;
        locate_batch(new_cnt, B, m, xyz_cnt, &start);
        row = start + idx[p];
        dx = xyz[(int64_t)row * 3 + 0] - new_xyz[(int64_t)m * 3 + 0];
        dy = xyz[(int64_t)row * 3 + 1] - new_xyz[(int64_t)m * 3 + 1];
        dz = xyz[(int64_t)row * 3 + 2] - new_xyz[(int64_t)m * 3 + 2];
      }
      if (rel) { rel[p * 3 + 0] = dx; rel[p * 3 + 1] = dy; rel[p * 3 + 2] = dz; }
    }
    srow[threadIdx.x] = row;
    sd[threadIdx.x][0] = dx; sd[threadIdx.x][1] = dy; sd[threadIdx.x][2] = dz;
  }
  __syncthreads();
  const int npl = (int)min((int64_t)64, MP - p0);
  if (HT > 0) {
    constexpr int H = HT > 0 ? HT : 4, H4 = H / 4, NPLANE = 256 / H4;
    float* dst = out + p0 * H;
    const int c4 = threadIdx.x % H4, plane = threadIdx.x / H4, c = c4 * 4;
    const gf4 w0 = *reinterpret_cast<const gf4*>(W1x + c), w1 = *reinterpret_cast<const gf4*>(W1x + H + c),
              w2 = *reinterpret_cast<const gf4*>(W1x + 2 * H + c);
    gf4 s1 = (gf4){0.f, 0.f, 0.f, 0.f}, s2 = (gf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 64 / NPLANE; ++i) {
      const int pl = plane + i * NPLANE;
      if (pl >= npl) break;
      const int row = srow[pl];
      gf4 v = (gf4){0.f, 0.f, 0.f, 0.f};                     // empty ball: the reference zeroes the whole grouped row
      if (row >= 0) {
        v = *reinterpret_cast<const gf4*>(P + (int64_t)row * H + c);
        const float dx = sd[pl][0], dy = sd[pl][1], dz = sd[pl][2];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaf(w2[k], dz, fmaf(w1[k], dy, fmaf(w0[k], dx, v[k])));
      }
      if (out) *reinterpret_cast<gf4*>(dst + pl * H + c) = v;    // out == NULL (with stat): statistics only (sa_mlp_train.hip pass 0)
      s1 += v;
      s2 += v * v;
    }
    if (stat) {                                              // wave-uniform
      __shared__ float sred[4][2][HT > 0 ? HT : 4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float a = s1[k], b = s2[k];
#pragma unroll
        for (int off = H4; off < 64; off <<= 1) {            // lanes of a wave with the same c4 sit H4 apart
          a += __shfl_xor(a, off);
          b += __shfl_xor(b, off);
        }
        s1[k] = a;
        s2[k] = b;
      }
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      if (lane < H4 && H4 <= 64) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          sred[wave][0][lane * 4 + k] = s1[k];
          sred[wave][1][lane * 4 + k] = s2[k];
        }
      }
      __syncthreads();
      // H4 <= 64: every wave holds all H4 column groups (waves differ in their planes); fixed order over the 4 waves
      for (int e = threadIdx.x; e < 2 * H; e += 256) {
        const int which = e / H, ch = e % H;
        float acc = 0.f;
        if (H4 <= 64) {
#pragma unroll
          for (int w = 0; w < 4; ++w) acc += sred[w][which][ch];
        }
        stat[((int64_t)blockIdx.x * 2 + which) * H + ch] = acc;
      }
    }
  } else {
    const int H = Hrt;
    float* dst = out + p0 * H;
    for (int e = threadIdx.x; e < npl * H; e += 256) {
      const int pl = e / H, c = e - pl * H;
      const int row = srow[pl];
      float v = 0.f;
      if (row >= 0)
        v = fmaf(W1x[2 * H + c], sd[pl][2], fmaf(W1x[H + c], sd[pl][1], fmaf(W1x[c], sd[pl][0], P[(int64_t)row * H + c])));
      dst[e] = v;
    }
  }
}

// backward of the kernel above: grad_P[row] += dy[p], and this slab's contribution to dW1x,
// part[blk][d][c] = sum_p rel[p][d] * dy[p][c] (summed over blocks by the caller).
// A ball query pads its result with its first hit, so most pairs of a 16-pair segment repeat an earlier pair's row: those
// are folded into the first occurrence inside LDS and only first occurrences issue global atomics, one row of H consecutive
// floats per H lanes (measured: the global atomics that remain cost < 0.01 ms at the RoI-grid shape; without any folding
// they cost 1.5 ms).
// H is a template parameter: with a run-time H the index arithmetic (e / H, e % H per element) made the kernel VALU-bound
// (~3000 instructions per thread, 2.4 ms at the RoI-grid shape for 1.8 GB of input).
// BN: grad_out is the gradient w.r.t. z = relu(batchnorm(y)) of the layer's output y (the rows `y_rows`) and the BatchNorm
// backward is applied while the slab is loaded — dy = gamma invstd (d - dbeta/n - xhat dgamma/n), d = dz [z > 0], exactly
// bn_bwd_apply_kernel's arithmetic — so the (M*nsample, H) gradient of y is never written and read back.
struct GroupBn {
  // RECOMP instances (crb_group_affine_rows_grad_bn_recompute_stack): y_rows / rel are not kept by the forward, the kernel forms
  // them again from xyz, new_xyz, P, W1x exactly like group_affine_rows_kernel
  const float* xyz;
  const float* new_xyz;
  const float* P;
  const float* W1x;
  // RECOMP instances, optional: the pairs in SOURCE-ROW order (crb_pair_sort_by_source: sorted_pair[k] = pair index, sorted_row[k] =
  // its source row, n_src for the pairs of empty balls, which sort to the end). A slab then holds runs of pairs with one target: the
  // fold below adds a run in LDS and issues ONE row of atomics per (16-pair segment, run) - at the voxel levels of the VSA module,
  // where every sample of a ball is a different row and many balls share a voxel, 5 - 8 x fewer atomics than in pair order
  const int* sorted_pair;
  const int* sorted_row;
  int n_src;
  const float* y_rows;
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
  const float* dbeta;
  const float* dgamma;
  float inv_n;
  // deterministic mode (crb_group_affine_rows_grad_bn_recompute_stack_fixed): the rows that leave a slab are added as 64-bit
  // fixed-point numbers (value * fixed_scale, rounded to nearest) - integer addition does not depend on the order of arrival
  unsigned long long* fixed;
  float fixed_scale;
};

template <int H, bool BN = false, bool RECOMP = false>
__global__ __launch_bounds__(256) void group_affine_rows_grad_kernel(int B, int64_t MP, int ns,
                                                                     const int* __restrict__ xyz_cnt,
                                                                     const int* __restrict__ new_cnt,
                                                                     const int* __restrict__ idx,
                                                                     const unsigned char* __restrict__ empty,
                                                                     const float* __restrict__ rel,
                                                                     const float* __restrict__ grad_out,
                                                                     float* __restrict__ grad_P, float* __restrict__ part,
                                                                     GroupBn bn = GroupBn{}) {
  constexpr int H4 = H / 4, NPLANE = 256 / H4, NROW = 256 / H < 1 ? 1 : 256 / H;
  __shared__ __attribute__((aligned(16))) float slab[64 * H];
  __shared__ float wred[4 * H4 * 12];
  __shared__ int srow[64];
  __shared__ int first[64];
  __shared__ float sd[64][3];
  const int64_t p0 = (int64_t)blockIdx.x * 64;
  int myrow = -1;
  __shared__ int spair[64];                                  // pair index of slot pl (identity order: p0 + pl)
  const bool sorted = RECOMP && bn.sorted_pair != nullptr;
  if (threadIdx.x < 64) {
    const int64_t k = p0 + threadIdx.x;
    int64_t p = k;
    int row = -1;
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (sorted) {
      if (k < MP) {
        p = bn.sorted_pair[k];
        const int rr = bn.sorted_row[k];
        if (rr < bn.n_src) {
          row = rr;
          myrow = row;
          const int m = (int)(p / ns);
          dx = bn.xyz[(int64_t)row * 3 + 0] - bn.new_xyz[(int64_t)m * 3 + 0];
          dy = bn.xyz[(int64_t)row * 3 + 1] - bn.new_xyz[(int64_t)m * 3 + 1];
          dz = bn.xyz[(int64_t)row * 3 + 2] - bn.new_xyz[(int64_t)m * 3 + 2];
        }
      }
      spair[threadIdx.x] = (int)p;
    } else if (p < MP) {
      spair[threadIdx.x] = (int)p;
      const int m = (int)(p / ns);
      if (!empty[m]) {
        int start;
        locate_batch(new_cnt, B, m, xyz_cnt, &start);
        row = start + idx[p];
        myrow = row;
        if constexpr (RECOMP) {
          dx = bn.xyz[(int64_t)row * 3 + 0] - bn.new_xyz[(int64_t)m * 3 + 0];
          dy = bn.xyz[(int64_t)row * 3 + 1] - bn.new_xyz[(int64_t)m * 3 + 1];
          dz = bn.xyz[(int64_t)row * 3 + 2] - bn.new_xyz[(int64_t)m * 3 + 2];
        } else {
          dx = rel[p * 3 + 0]; dy = rel[p * 3 + 1]; dz = rel[p * 3 + 2];
        }
      }
    }
    srow[threadIdx.x] = row;
    sd[threadIdx.x][0] = dx; sd[threadIdx.x][1] = dy; sd[threadIdx.x][2] = dz;
  }
  if constexpr (RECOMP) {
    // the slab loads below need srow / sd. A slab of empty balls only has nothing to scatter and rel = 0, and its producer
    // (sa_mlp_train.hip pass C) did not write its rows: zero part, done
    if (!__syncthreads_or(myrow >= 0)) {
      for (int e = threadIdx.x; e < 3 * H; e += 256) part[(int64_t)blockIdx.x * 3 * H + e] = 0.f;
      return;
    }
  }
  const int npl = (int)min((int64_t)64, MP - p0);
  const int c4 = threadIdx.x % H4, plane = threadIdx.x / H4;
  const gf4* src = reinterpret_cast<const gf4*>(grad_out + p0 * H);
  gf4* slab4 = reinterpret_cast<gf4*>(slab);
  gf4 v[64 / NPLANE];
  gf4 mu, is, ga, be, db, dg;
  if constexpr (BN) {
    mu = *reinterpret_cast<const gf4*>(bn.mean + 4 * c4);
    is = *reinterpret_cast<const gf4*>(bn.invstd + 4 * c4);
    ga = *reinterpret_cast<const gf4*>(bn.gamma + 4 * c4);
    be = *reinterpret_cast<const gf4*>(bn.beta + 4 * c4);
    db = *reinterpret_cast<const gf4*>(bn.dbeta + 4 * c4);
    dg = *reinterpret_cast<const gf4*>(bn.dgamma + 4 * c4);
  }
#pragma unroll
  for (int i = 0; i < 64 / NPLANE; ++i) {
    const int pl = plane + i * NPLANE;
    bool have = pl < npl;
    if constexpr (RECOMP) have = have && srow[pl] >= 0;      // rows of empty balls were not written by the producer
    if (sorted) v[i] = have ? reinterpret_cast<const gf4*>(grad_out + (int64_t)spair[pl] * H)[c4] : (gf4){0.f, 0.f, 0.f, 0.f};
    else v[i] = have ? src[pl * H4 + c4] : (gf4){0.f, 0.f, 0.f, 0.f};
    if constexpr (BN) {
      if (have) {
        gf4 yv;
        if constexpr (RECOMP) {
          yv = (gf4){0.f, 0.f, 0.f, 0.f};
          const int row = srow[pl];
          if (row >= 0) {
            yv = *reinterpret_cast<const gf4*>(bn.P + (int64_t)row * H + 4 * c4);
            const gf4 w0 = *reinterpret_cast<const gf4*>(bn.W1x + 4 * c4), w1 = *reinterpret_cast<const gf4*>(bn.W1x + H + 4 * c4),
                      w2 = *reinterpret_cast<const gf4*>(bn.W1x + 2 * H + 4 * c4);
            const float dx = sd[pl][0], dy = sd[pl][1], dz = sd[pl][2];
#pragma unroll
            for (int k = 0; k < 4; ++k) yv[k] = fmaf(w2[k], dz, fmaf(w1[k], dy, fmaf(w0[k], dx, yv[k])));
          }
        } else {
          yv = reinterpret_cast<const gf4*>(bn.y_rows + p0 * H)[pl * H4 + c4];
        }
        const gf4 xh = (yv - mu) * is;
        const gf4 z = ga * xh + be;
        gf4 d = v[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = z[k] > 0.f ? d[k] : 0.f;
        v[i] = ga * is * (d - db * bn.inv_n - xh * (dg * bn.inv_n));
      }
    }
    slab4[pl * H4 + c4] = v[i];
  }
  __syncthreads();
  {                                                          // dW1x piece from the registers (rel = 0 on empty rows)
    float wacc[3][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < 64 / NPLANE; ++i) {
      const int pl = plane + i * NPLANE;
      const float dx = sd[pl][0], dy = sd[pl][1], dz = sd[pl][2];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        wacc[0][k] = fmaf(dx, v[i][k], wacc[0][k]);
        wacc[1][k] = fmaf(dy, v[i][k], wacc[1][k]);
        wacc[2][k] = fmaf(dz, v[i][k], wacc[2][k]);
      }
    }
    // lanes of a wave that share c4 sit H4 apart: butterfly over them, then one (4 waves x H4 x 12) LDS table
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float t = wacc[d][k];
#pragma unroll
        for (int off = H4; off < 64; off <<= 1) t += __shfl_xor(t, off);
        wacc[d][k] = t;
      }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < H4) {
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int k = 0; k < 4; ++k) wred[(wave * H4 + lane) * 12 + d * 4 + k] = wacc[d][k];
    }
  }
  if (threadIdx.x < 64) {                                    // first occurrence of this pair's row in its 16-pair segment
    const int row = srow[threadIdx.x];
    int f = threadIdx.x;
    if (sorted) {
      // source-row order: equal rows are adjacent, the first of a run inside the segment = the highest run start at or below me
      const bool startrun = (threadIdx.x & 15) == 0 || srow[threadIdx.x - 1] != row;
      const unsigned long long starts = __ballot(startrun);              // (threads 0..63 = wave 0)
      const unsigned long long below = starts & ((2ULL << threadIdx.x) - 1ULL);
      f = 63 - __builtin_clzll(below);
    } else if (row >= 0) {
      for (int q = threadIdx.x & ~15; q < (int)threadIdx.x; ++q)
        if (srow[q] == row) { f = q; break; }
    }
    first[threadIdx.x] = f;
  }
  __syncthreads();
  if (threadIdx.x < 3 * H) {
    const int d = threadIdx.x / H, c = threadIdx.x % H;
    float acc = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) acc += wred[(w * H4 + (c >> 2)) * 12 + d * 4 + (c & 3)];
    part[(int64_t)blockIdx.x * 3 * H + threadIdx.x] = acc;
  }
  if (3 * H > 256) {                                         // H = 128: 384 outputs
    for (int e = 256 + threadIdx.x; e < 3 * H; e += 256) {
      const int d = e / H, c = e % H;
      float acc = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) acc += wred[(w * H4 + (c >> 2)) * 12 + d * 4 + (c & 3)];
      part[(int64_t)blockIdx.x * 3 * H + e] = acc;
    }
  }
  // fold repeats into their first occurrence: one thread per (segment, channel) walks its 16 pairs; runs of pairs with the
  // same target (the padded tail of a ball) are summed in a register. (ds_add_f32 from all lanes instead: +1.3 ms at the
  // RoI-grid shape — the repeats of one ball hit the same addresses from every wave.)
  for (int e = threadIdx.x; e < 4 * H; e += 256) {
    const int seg = e / H, c = e % H;
    int cur = -1;
    float acc = 0.f;
#pragma unroll 4
    for (int pl = 16 * seg; pl < 16 * seg + 16; ++pl) {
      const int f = first[pl];
      if (f == pl) continue;
      if (f != cur) {
        if (cur >= 0) slab[cur * H + c] += acc;
        cur = f;
        acc = 0.f;
      }
      acc += slab[pl * H + c];
    }
    if (cur >= 0) slab[cur * H + c] += acc;
  }
  __syncthreads();
  const int c = threadIdx.x % H, r0 = threadIdx.x / H;
  if (sorted) {
    // source-row order: a run goes on across the segment borders (a keypoint of the RoI-grid scales has hundreds of pairs). The
    // head of a segment whose row continues the previous segment's last run is added to that run's head, last segment first, so a
    // run over several whole segments arrives at its first head
    if (threadIdx.x < H) {
#pragma unroll
      for (int seg = 3; seg >= 1; --seg) {
        const int h = 16 * seg, row = srow[h];
        if (row >= 0 && srow[h - 1] == row) slab[first[h - 1] * H + c] += slab[h * H + c];
      }
    }
    __syncthreads();
  }
  // one source row per H consecutive lanes
  for (int pl = r0; pl < 64; pl += NROW) {
    const int row = srow[pl];
    bool head = row >= 0 && first[pl] == pl;
    if (sorted && pl > 0 && (pl & 15) == 0 && srow[pl - 1] == row) head = false;
    if (head) {
      const float val = slab[pl * H + c];
      if (bn.fixed) atomicAdd(&bn.fixed[(int64_t)row * H + c], (unsigned long long)__float2ll_rn(val * bn.fixed_scale));
      else atomicAdd(&grad_P[(int64_t)row * H + c], val);
    }
  }
}

// ------------------------------------------------------------------------------------------------ FPS
// Tie rule of the reference (sampling_gpu.cu:49-139) as a strict total order on candidates k:
//   larger running distance first; then smaller bit-reversed (k mod bs) (its LDS tree keeps the lower-position operand
//   on ties); then smaller k (each strided thread keeps its first maximum). bs = min(2^floor(log2 n), 1024).
constexpr int FPS_THREADS = 1024;
constexpr int FPS_REG_MAX_PER_THREAD = 40;    // up to 40960 points per frame with register-resident distances

// REGS: point coordinates live in registers for the whole run (n <= 20480); otherwise they are re-read (coalesced, L2
// resident) every round and only the running distances stay in registers. The tie key is built once per thread per
// round from its best index (a per-point key array spilled the 128-VGPR budget of a 1024-thread workgroup).
template <int PPT, bool REGS>
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(int n, int m, int log2bs, const float* __restrict__ xyz_all,
                                                          int* __restrict__ out_all) {
  __shared__ unsigned long long wave_best[16];
  __shared__ float sel_xyz[3];
  const float* xyz = xyz_all + (int64_t)blockIdx.x * n * 3;
  int* out = out_all + (int64_t)blockIdx.x * m;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float px[REGS ? PPT : 1], py[REGS ? PPT : 1], pz[REGS ? PPT : 1], dist[PPT];
  const unsigned bsmask = (1u << log2bs) - 1u;
#pragma unroll
  for (int t = 0; t < PPT; ++t) {
    const int k = tid + t * FPS_THREADS;
    if (REGS) {
      const bool in = k < n;
      px[t] = in ? xyz[k * 3 + 0] : 0.f; py[t] = in ? xyz[k * 3 + 1] : 0.f; pz[t] = in ? xyz[k * 3 + 2] : 0.f;
    }
    dist[t] = 1e10f;
  }
  if (tid == 0) { out[0] = 0; sel_xyz[0] = xyz[0]; sel_xyz[1] = xyz[1]; sel_xyz[2] = xyz[2]; }
  __syncthreads();
  for (int j = 1; j < m; ++j) {
    const float x1 = sel_xyz[0], y1 = sel_xyz[1], z1 = sel_xyz[2];
    float bestd = -1.f;
    int bestk = 0;
#pragma unroll
    for (int t = 0; t < PPT; ++t) {
      const int k = tid + t * FPS_THREADS;
      if (k < n) {
        float x2, y2, z2;
        if (REGS) { x2 = px[t]; y2 = py[t]; z2 = pz[t]; }
        else { x2 = xyz[k * 3 + 0]; y2 = xyz[k * 3 + 1]; z2 = xyz[k * 3 + 2]; }
        const float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
        const float d = dx * dx + dy * dy + dz * dz;
        const float d2 = fminf(d, dist[t]);
        dist[t] = d2;
        if (d2 > bestd) { bestd = d2; bestk = k; }     // strict: the first maximum of the strided thread wins
      }
    }
    unsigned long long best = 0ULL;
    if (bestd >= 0.f) {
      const unsigned v = (unsigned)bestk & bsmask;
      const unsigned br = log2bs ? (__brev(v) >> (32 - log2bs)) : 0u;
      best = ((unsigned long long)__float_as_uint(bestd) << 32) | (0xffffffffu - ((br << 20) | (unsigned)bestk));
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
      unsigned long long o = __shfl_xor(best, s, 64);
      best = o > best ? o : best;
    }
    if (lane == 0) wave_best[wave] = best;
    __syncthreads();
    unsigned long long b2 = wave_best[lane & 15];
#pragma unroll
    for (int s = 8; s > 0; s >>= 1) {
      unsigned long long o = __shfl_xor(b2, s, 64);
      b2 = o > b2 ? o : b2;
    }
    const int sel = (int)((0xffffffffu - (unsigned)(b2 & 0xffffffffULL)) & 0xfffffu);
    if (tid == 0) {
      out[j] = sel;
      sel_xyz[0] = xyz[sel * 3 + 0]; sel_xyz[1] = xyz[sel * 3 + 1]; sel_xyz[2] = xyz[sel * 3 + 2];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ three NN / interp
__global__ __launch_bounds__(256) void three_nn_kernel(int B, int N, const float* __restrict__ unknown,
                                                       const int* __restrict__ unknown_cnt,
                                                       const float* __restrict__ known, const int* __restrict__ known_cnt,
                                                       float* __restrict__ dist2, int* __restrict__ idx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  int start;
  const int b = locate_batch(unknown_cnt, B, i, known_cnt, &start);
  const int n = known_cnt[b];
  const float ux = unknown[(int64_t)i * 3 + 0], uy = unknown[(int64_t)i * 3 + 1], uz = unknown[(int64_t)i * 3 + 2];
  const float* kp = known + (int64_t)start * 3;
  double b1 = 1e40, b2 = 1e40, b3 = 1e40;      // the reference compares in double (interpolate_gpu.cu:51)
  int i1 = 0, i2 = 0, i3 = 0;
  for (int k = 0; k < n; ++k) {
    float x = kp[k * 3 + 0], y = kp[k * 3 + 1], z = kp[k * 3 + 2];
    float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
    if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; }
    else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = k; }
    else if (d < b3) { b3 = d; i3 = k; }
  }
  dist2[(int64_t)i * 3 + 0] = (float)b1; dist2[(int64_t)i * 3 + 1] = (float)b2; dist2[(int64_t)i * 3 + 2] = (float)b3;
  idx[(int64_t)i * 3 + 0] = i1 + start; idx[(int64_t)i * 3 + 1] = i2 + start; idx[(int64_t)i * 3 + 2] = i3 + start;
}

__global__ __launch_bounds__(256) void three_interp_kernel(int N, int C, const float* __restrict__ feat,
                                                           const int* __restrict__ idx, const float* __restrict__ w,
                                                           float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)N * C) return;
  const int i = (int)(t / C), c = (int)(t - (int64_t)i * C);
  const int* id = idx + (int64_t)i * 3;
  const float* ww = w + (int64_t)i * 3;
  out[t] = ww[0] * feat[(int64_t)id[0] * C + c] + ww[1] * feat[(int64_t)id[1] * C + c] +
           ww[2] * feat[(int64_t)id[2] * C + c];
}

__global__ __launch_bounds__(256) void three_interp_grad_kernel(int N, int C, const float* __restrict__ grad_out,
                                                                const int* __restrict__ idx, const float* __restrict__ w,
                                                                float* __restrict__ grad_feat) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)N * C) return;
  const int i = (int)(t / C), c = (int)(t - (int64_t)i * C);
  const int* id = idx + (int64_t)i * 3;
  const float* ww = w + (int64_t)i * 3;
  const float g = grad_out[t];
  atomicAdd(&grad_feat[(int64_t)id[0] * C + c], g * ww[0]);
  atomicAdd(&grad_feat[(int64_t)id[1] * C + c], g * ww[1]);
  atomicAdd(&grad_feat[(int64_t)id[2] * C + c], g * ww[2]);
}


// ------------------------------------------------------------------------------------------------ ball query on a cell grid
// (round 6) The scans above test every point of the frame against every query
// (ball_query_gpu.cu:47-66 does the same per thread). Here the points of one call are put into a hashed grid of cells of edge
// 1.001 r_b by a counting sort of the call's own (histogram with atomics, exclusive scan, cursor fill: 5 launches, no library
// sort), a query reads the 27 cells around its own (distinct hash buckets only), tests those candidates with the SAME distance
// expression, and keeps the nsample SMALLEST point indices of each radius in ascending order: the list a scan in index order would
// have produced (ball_query_gpu.cu:49-64: hits in scan order, first nsample, padded with the first hit), index-exact. A bucket may
// hold points of other cells or frames (hash collisions): they fail the frame or the distance test like any other point. Queries
// with more than BQG_MAX_CAND candidates or BQG_MAX_HITS hits in r_b (dense balls) are answered by the scan, which ends early there.
// Why 1.001: a point with |dx| < r_b has to lie in a neighbouring cell; floor(x / c) of the two sides is computed in f32 (error <= 3e-5
// cells at 250 cells from the origin), the margin keeps the difference of the floors <= 1.
constexpr int BQG_MAX_CAND = 4096, BQG_MAX_HITS = 1024;

struct BqGrid {
  float inv_cell;
  unsigned mask;       // buckets - 1 (power of two)
};
__device__ __forceinline__ unsigned bqg_bucket(const BqGrid& g, int b, int ix, int iy, int iz) {
  unsigned h = (unsigned)ix * 73856093u ^ (unsigned)iy * 19349663u ^ (unsigned)iz * 83492791u ^ (unsigned)b * 2654435761u;
  h ^= h >> 15;
  return h & g.mask;
}
// exclusive prefixes of the two count vectors, B + 1 entries each (one wave; B is a batch size)
__global__ __launch_bounds__(64) void bqg_prefix_kernel(int B, const int* __restrict__ xyz_cnt, const int* __restrict__ new_cnt,
                                                        int* __restrict__ xyz_pref, int* __restrict__ new_pref) {
  int ax = 0, an = 0;
  for (int k0 = 0; k0 < B; k0 += 64) {
    const int k = k0 + (int)threadIdx.x;
    const int cx = k < B ? xyz_cnt[k] : 0, cn = k < B ? new_cnt[k] : 0;
    const int ix = crb_wave_incl_scan(cx), in = crb_wave_incl_scan(cn);
    if (k < B) { xyz_pref[k] = ax + ix - cx; new_pref[k] = an + in - cn; }
    ax += __shfl(ix, 63, 64);
    an += __shfl(in, 63, 64);
  }
  if (threadIdx.x == 0) { xyz_pref[B] = ax; new_pref[B] = an; }
}
// frame of row i: the last b with pref[b] <= i (rows past the end: the last frame)
__device__ __forceinline__ int bqg_frame_of(const int* __restrict__ pref, int B, int i) {
  int lo = 0, hi = B - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (pref[mid] <= i) lo = mid; else hi = mid - 1;
  }
  return lo;
}
__global__ __launch_bounds__(256) void bqg_hist_kernel(BqGrid g, int B, int n, const float* __restrict__ xyz, const int* __restrict__ xyz_pref,
                                                       int* __restrict__ count, int* __restrict__ bucket_of) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int b = bqg_frame_of(xyz_pref, B, i);
  const unsigned h = bqg_bucket(g, b, (int)floorf(xyz[(int64_t)i * 3] * g.inv_cell), (int)floorf(xyz[(int64_t)i * 3 + 1] * g.inv_cell),
                                (int)floorf(xyz[(int64_t)i * 3 + 2] * g.inv_cell));
  bucket_of[i] = (int)h;
  atomicAdd(&count[h], 1);
}
__global__ __launch_bounds__(256) void bqg_fill_kernel(int n, const int* __restrict__ bucket_of, int* __restrict__ cursor, int* __restrict__ sorted) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  sorted[atomicAdd(&cursor[bucket_of[i]], 1)] = i;      // (order inside a bucket: arbitrary; the query kernel orders its hits by index)
}

__global__ __launch_bounds__(256) void ball_query2_grid_kernel(BqGrid g, int B, int M, float ra, int nsa, float rb, int nsb,
                                                               const float* __restrict__ new_xyz, const int* __restrict__ new_cnt,
                                                               const float* __restrict__ xyz, const int* __restrict__ xyz_cnt,
                                                               const int* __restrict__ xyz_pref, const int* __restrict__ new_pref,
                                                               const int* __restrict__ bstart, const int* __restrict__ sorted,
                                                               int* __restrict__ idx_a, int* __restrict__ idx_b,
                                                               unsigned char* __restrict__ empty_a, unsigned char* __restrict__ empty_b) {
  __shared__ int s_pref[4][64], s_beg[4][32];           // (s_pref doubles as the selection scratch: nsample <= 64)
  __shared__ int s_hit[4][BQG_MAX_HITS];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + wave;
  if (q >= M) return;
  const int b = bqg_frame_of(new_pref, B, q);               // (log2 B cached loads: locate_batch's B dependent loads per wave cost more than the query)
  const int start = xyz_pref[b];
  const int n = xyz_pref[b + 1] - start;
  const float* p = xyz + (int64_t)start * 3;
  const float qx = new_xyz[(int64_t)q * 3 + 0], qy = new_xyz[(int64_t)q * 3 + 1], qz = new_xyz[(int64_t)q * 3 + 2];
  const float ra2 = ra * ra, rb2 = rb * rb;
  const unsigned long long below = (1ULL << lane) - 1ULL;
  int* oa = idx_a + (int64_t)q * nsa;
  int* ob = idx_b + (int64_t)q * nsb;
  // ---- the 27 buckets around the query's cell, duplicates dropped
  const int cx = (int)floorf(qx * g.inv_cell), cy = (int)floorf(qy * g.inv_cell), cz = (int)floorf(qz * g.inv_cell);
  int hb = -1, beg = 0, len = 0;
  if (lane < 27) {
    hb = (int)bqg_bucket(g, b, cx + lane % 3 - 1, cy + (lane / 3) % 3 - 1, cz + lane / 9 - 1);
    beg = bstart[hb];
    len = bstart[hb + 1] - beg;
  }
  for (int j = 0; j < 26; ++j) {
    const int hj = __builtin_amdgcn_readlane(hb, j);
    if (lane > j && lane < 27 && hb == hj) len = 0;
  }
  const int pref = crb_wave_incl_scan(len) - len;                          // candidates in front of this lane's bucket
  const int T = __builtin_amdgcn_readlane(pref + len, 63);
  bool scan = T > BQG_MAX_CAND;
  int H = 0;
  if (!scan) {
    if (lane < 32) { s_pref[wave][lane] = lane < 27 ? pref : 0x7fffffff; s_beg[wave][lane] = beg; }
    __builtin_amdgcn_wave_barrier();
    // (wave-private LDS, in-order LDS pipe: the stores above are visible to this wave's reads below)
    for (int c0 = 0; c0 < T; c0 += 256) {                                  // four candidates per lane and step: four load chains in flight
      int k[4];
      float d2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = c0 + u * 64 + lane;
        k[u] = -1;
        if (t < T) {
          int lo = 0;                                                      // last bucket whose prefix is <= t (empty buckets share a prefix: take the last)
#pragma unroll
          for (int s = 16; s > 0; s >>= 1)
            if (lo + s < 27 && s_pref[wave][lo + s] <= t) lo += s;
          k[u] = sorted[s_beg[wave][lo] + (t - s_pref[wave][lo])] - start;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        d2[u] = __builtin_inff();
        if (k[u] >= 0 && k[u] < n) {
          const float x = p[k[u] * 3 + 0], y = p[k[u] * 3 + 1], z = p[k[u] * 3 + 2];
          d2[u] = (qx - x) * (qx - x) + (qy - y) * (qy - y) + (qz - z) * (qz - z);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool hit_b = d2[u] < rb2, hit_a = d2[u] < ra2;
        const unsigned long long mb = __ballot(hit_b);
        if (mb != 0ULL) {
          const int pos = H + __popcll(mb & below);
          if (hit_b && pos < BQG_MAX_HITS) s_hit[wave][pos] = k[u] | (hit_a ? (int)0x80000000 : 0);
          H += __popcll(mb);
        }
      }
    }
    scan = H > BQG_MAX_HITS;
    __builtin_amdgcn_wave_barrier();
  }
  if (scan) {                                                               // dense ball: the scan in index order ends after a few points
    int ca = 0, cb = 0, fa = -1, fb = -1;
    for (int k0 = 0; k0 < n && (ca < nsa || cb < nsb); k0 += 64) {
      const int k = k0 + lane;
      bool ha = false, hbb = false;
      if (k < n) {
        const float x = p[k * 3 + 0], y = p[k * 3 + 1], z = p[k * 3 + 2];
        const float d2 = (qx - x) * (qx - x) + (qy - y) * (qy - y) + (qz - z) * (qz - z);
        ha = d2 < ra2;
        hbb = d2 < rb2;
      }
      const unsigned long long ma = __ballot(ha), mb = __ballot(hbb);
      if (ma != 0ULL && ca < nsa) {
        if (fa < 0) fa = k0 + (__ffsll((long long)ma) - 1);
        const int pos = ca + __popcll(ma & below);
        if (ha && pos < nsa) oa[pos] = k;
        ca += __popcll(ma);
      }
      if (mb != 0ULL && cb < nsb) {
        if (fb < 0) fb = k0 + (__ffsll((long long)mb) - 1);
        const int pos = cb + __popcll(mb & below);
        if (hbb && pos < nsb) ob[pos] = k;
        cb += __popcll(mb);
      }
    }
    ca = ca > nsa ? nsa : ca;
    cb = cb > nsb ? nsb : cb;
    for (int l = ca + lane; l < nsa; l += 64) oa[l] = (fa >= 0) ? fa : 0;
    for (int l = cb + lane; l < nsb; l += 64) ob[l] = (fb >= 0) ? fb : 0;
    if (lane == 0) { empty_a[q] = fa < 0; empty_b[q] = fb < 0; }
    return;
  }
  // ---- the nsample smallest indices of each radius, in ascending order (H <= BQG_MAX_HITS hits in s_hit, bit 31: inside r_a too)
  auto emit = [&](bool want_a, int ns, int* out, unsigned char* empty_flag) {
    // count of the radius, and the largest index that is still among the ns smallest (threshold search over the index range)
    int cnt = 0;
    for (int i0 = 0; i0 < H; i0 += 64) {
      const int i = i0 + lane;
      const bool in = i < H && (!want_a || s_hit[wave][i] < 0);
      cnt += __popcll(__ballot(in));
    }
    int thr = 0x7fffffff;
    if (cnt > ns) {
      int lo = 0, hi = n - 1;                                               // smallest v with #(k <= v) >= ns
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        int c = 0;
        for (int i0 = 0; i0 < H; i0 += 64) {
          const int i = i0 + lane;
          const int v = i < H ? s_hit[wave][i] : 0;
          c += __popcll(__ballot(i < H && (!want_a || v < 0) && (v & 0x7fffffff) <= mid));
        }
        if (c >= ns) hi = mid; else lo = mid + 1;
      }
      thr = lo;
    }
    const int total = cnt < ns ? cnt : ns;
    // the selected hits (at most ns <= 64 of them) compacted into the wave's bucket scratch, then ranked among themselves
    int* sel = s_pref[wave];                                                // (s_pref / s_beg: 64 ints, done with)
    int at = 0;
    for (int i0 = 0; i0 < H; i0 += 64) {
      const int i = i0 + lane;
      const int v = i < H ? s_hit[wave][i] : 0;
      const bool pick = i < H && (!want_a || v < 0) && (v & 0x7fffffff) <= thr;
      const unsigned long long m = __ballot(pick);
      if (pick) sel[at + __popcll(m & below)] = v & 0x7fffffff;
      at += __popcll(m);
    }
    __builtin_amdgcn_wave_barrier();
    int mine = 0x7fffffff, rank = 0;
    if (lane < total) {
      mine = sel[lane];
      for (int j = 0; j < total; ++j) rank += sel[j] < mine ? 1 : 0;
      out[rank] = mine;
    }
    int first = mine;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) first = min(first, __shfl_xor(first, s, 64));
    for (int l = total + lane; l < ns; l += 64) out[l] = total > 0 ? first : 0;
    if (lane == 0) *empty_flag = total == 0;
    __builtin_amdgcn_wave_barrier();
  };
  emit(true, nsa, oa, empty_a + q);
  emit(false, nsb, ob, empty_b + q);
}

// ---- round 6: the in-register sampling kernel again, built for fewer vector instructions and ONE barrier per
// round. fps_kernel spends a round (2.6 us at 20,000 points) on ~240 vector instructions per wave for the 20 distance updates with
// their running (distance, index) pair, a 64-bit butterfly through ds_bpermute (20 of them), two barriers and a dependent global
// load of the winner's coordinates by thread 0. Here:
//   * the update runs on PAIRS of points in packed f32 (v_pk_add / v_pk_mul: the same IEEE operations in the
//     same order, two points per instruction; min / max have no packed form) and keeps only the running maximum; the index of a lane's maximum is found
//     afterwards (first t with dist[t] == maximum: the strided thread's first maximum, as before);
//   * wave level: maximum distance by DPP row shifts + row broadcasts (6 steps, no LDS), lanes that hold it offer the tie key
//     ((bit-reversed index bits << 20) | index, smallest wins: the order of the 64-bit keys of fps_kernel), minimum by DPP again;
//     the winning lane fetches its point's coordinates with one 12-byte load (the cloud is L2-resident: 16 loads per round in
//     parallel, one per wave, instead of thread 0's load behind the second barrier);
//   * workgroup level: the winner lane of every wave writes {distance, key, x, y, z} into a double-buffered slot, ONE barrier, every
//     wave reduces the 16 slots inside a DPP row and reads the winner's coordinates from the slot: no second barrier.
// Picks are the picks of fps_kernel (tests/test_pointnet2_gpu.py: bit-exact against oracle_fps incl. the tie rule).
typedef float fps_f2 __attribute__((ext_vector_type(2)));

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float fps_dpp_f(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ unsigned fps_dpp_u(unsigned old, unsigned src) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, 0xf, false);
}
// maximum over the wave (lane 63 holds it after the two broadcasts; returned as a uniform value)
__device__ __forceinline__ float fps_wave_max(float v) {
  v = fmaxf(v, fps_dpp_f<0x111>(v, v));          // row_shr:1
  v = fmaxf(v, fps_dpp_f<0x112>(v, v));          // row_shr:2
  v = fmaxf(v, fps_dpp_f<0x114>(v, v));          // row_shr:4
  v = fmaxf(v, fps_dpp_f<0x118>(v, v));          // row_shr:8   -> lane 15 of every row: the row's maximum
  v = fmaxf(v, fps_dpp_f<0x142, 0xa>(v, v));     // row_bcast:15 into rows 1, 3
  v = fmaxf(v, fps_dpp_f<0x143, 0xc>(v, v));     // row_bcast:31 into rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ unsigned fps_wave_min_u(unsigned v) {
  v = min(v, fps_dpp_u<0x111>(v, v));
  v = min(v, fps_dpp_u<0x112>(v, v));
  v = min(v, fps_dpp_u<0x114>(v, v));
  v = min(v, fps_dpp_u<0x118>(v, v));
  v = min(v, fps_dpp_u<0x142, 0xa>(v, v));
  v = min(v, fps_dpp_u<0x143, 0xc>(v, v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

template <int PPT>      // points per thread, even
__global__ __launch_bounds__(FPS_THREADS) void fps2_kernel(int n, int m, int log2bs, const float* __restrict__ xyz_all,
                                                           int* __restrict__ out_all) {
  __shared__ float slot[2][16][8];                 // [buffer][wave]: distance, key (bits), x, y, z
  const float* xyz = xyz_all + (int64_t)blockIdx.x * n * 3;
  int* out = out_all + (int64_t)blockIdx.x * m;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  fps_f2 px[PPT / 2], py[PPT / 2], pz[PPT / 2], dist[PPT / 2];      // element e of pair h: point tid + (2 h + e) * 1024
  const unsigned bsmask = (1u << log2bs) - 1u;
#pragma unroll
  for (int h = 0; h < PPT / 2; ++h)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k = tid + (2 * h + e) * FPS_THREADS;
      const bool in = k < n;
      px[h][e] = in ? xyz[k * 3 + 0] : 0.f; py[h][e] = in ? xyz[k * 3 + 1] : 0.f; pz[h][e] = in ? xyz[k * 3 + 2] : 0.f;
      dist[h][e] = in ? 1e10f : -1.f;              // a slot past the end never wins: min(d, -1) = -1 < every real distance
    }
  float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];
  if (tid == 0) out[0] = 0;
  for (int j = 1; j < m; ++j) {
    const fps_f2 X1 = (fps_f2){x1, x1}, Y1 = (fps_f2){y1, y1}, Z1 = (fps_f2){z1, z1};
    float lmax = -1.f;
#pragma unroll
    for (int h = 0; h < PPT / 2; ++h) {
      const fps_f2 dx = px[h] - X1, dy = py[h] - Y1, dz = pz[h] - Z1;
      const fps_f2 d = dx * dx + dy * dy + dz * dz;
      // (no NaNs here: fminf / fmaxf add a canonicalising v_max x, x, x per operand; gfx950 has no packed f32 min / max)
      float d0, d1;
      asm("v_min_f32 %0, %1, %2" : "=v"(d0) : "v"(d[0]), "v"(dist[h][0]));
      asm("v_min_f32 %0, %1, %2" : "=v"(d1) : "v"(d[1]), "v"(dist[h][1]));
      dist[h] = (fps_f2){d0, d1};
      asm("v_max3_f32 %0, %1, %2, %3" : "=v"(lmax) : "v"(lmax), "v"(d0), "v"(d1));
    }
    const float bestd = lmax;
    const float M = fps_wave_max(bestd);
    // the lane's first maximum (descending scan: the smallest t is written last)
    int bestt = 0;
#pragma unroll
    for (int t = PPT - 1; t >= 0; --t) bestt = dist[t >> 1][t & 1] == bestd ? t : bestt;
    const int bestk = tid + bestt * FPS_THREADS;
    const unsigned v = (unsigned)bestk & bsmask;
    const unsigned br = log2bs ? (__brev(v) >> (32 - log2bs)) : 0u;
    const unsigned key = (bestd == M && bestd >= 0.f) ? ((br << 20) | (unsigned)bestk) : 0xffffffffu;
    const unsigned K = fps_wave_min_u(key);
    // the wave's winner lane fetches its point (one 12-byte load, L2-resident cloud) and fills the wave's slot
    float* sl = slot[j & 1][0];
    if (key == K && K != 0xffffffffu) {
      const float* p = xyz + (int64_t)bestk * 3;
      float* o = sl + wave * 8;
      o[0] = M; o[1] = __uint_as_float(K); o[2] = p[0]; o[3] = p[1]; o[4] = p[2];
    } else if (K == 0xffffffffu && lane == 0) {
      float* o = sl + wave * 8;
      o[0] = -1.f; o[1] = __uint_as_float(K);
    }
    __syncthreads();
    // every wave: the best of the 16 slots (inside one DPP row), then the winner's coordinates from its slot
    const int w16 = lane & 15;
    const float sd = sl[w16 * 8];
    const unsigned sk = __float_as_uint(sl[w16 * 8 + 1]);
    float gm = sd;
    gm = fmaxf(gm, fps_dpp_f<0x111>(gm, gm));
    gm = fmaxf(gm, fps_dpp_f<0x112>(gm, gm));
    gm = fmaxf(gm, fps_dpp_f<0x114>(gm, gm));
    gm = fmaxf(gm, fps_dpp_f<0x118>(gm, gm));
    const float GM = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gm), 15));
    unsigned gk = sd == GM ? sk : 0xffffffffu;
    gk = min(gk, fps_dpp_u<0x111>(gk, gk));
    gk = min(gk, fps_dpp_u<0x112>(gk, gk));
    gk = min(gk, fps_dpp_u<0x114>(gk, gk));
    gk = min(gk, fps_dpp_u<0x118>(gk, gk));
    const unsigned GK = (unsigned)__builtin_amdgcn_readlane((int)gk, 15);
    const unsigned long long wb = __ballot(sd == GM && sk == GK) & 0xffffULL;
    const int ww = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(wb ? wb : 1ULL));
    x1 = sl[ww * 8 + 2]; y1 = sl[ww * 8 + 3]; z1 = sl[ww * 8 + 4];
    if (tid == 0) out[j] = (int)(GK & 0xfffffu);
  }
}

// any n: running distances in a caller-provided (B,n) buffer, coordinates re-read every round
__global__ __launch_bounds__(FPS_THREADS) void fps_large_kernel(int n, int m, int log2bs, const float* __restrict__ xyz_all,
                                                                float* __restrict__ temp_all, int* __restrict__ out_all) {
  __shared__ unsigned long long wave_best[16];
  __shared__ float sel_xyz[3];
  const float* xyz = xyz_all + (int64_t)blockIdx.x * n * 3;
  float* dist = temp_all + (int64_t)blockIdx.x * n;
  int* out = out_all + (int64_t)blockIdx.x * m;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned bsmask = (1u << log2bs) - 1u;
  for (int k = tid; k < n; k += FPS_THREADS) dist[k] = 1e10f;
  if (tid == 0) { out[0] = 0; sel_xyz[0] = xyz[0]; sel_xyz[1] = xyz[1]; sel_xyz[2] = xyz[2]; }
  __syncthreads();
  for (int j = 1; j < m; ++j) {
    const float x1 = sel_xyz[0], y1 = sel_xyz[1], z1 = sel_xyz[2];
    float bestd = -1.f;
    int bestk = 0;
    for (int k = tid; k < n; k += FPS_THREADS) {
      const float dx = xyz[k * 3 + 0] - x1, dy = xyz[k * 3 + 1] - y1, dz = xyz[k * 3 + 2] - z1;
      const float d2 = fminf(dx * dx + dy * dy + dz * dz, dist[k]);
      dist[k] = d2;
      if (d2 > bestd) { bestd = d2; bestk = k; }
    }
    unsigned long long best = 0ULL;
    if (bestd >= 0.f) {
      const unsigned v = (unsigned)bestk & bsmask;
      const unsigned br = log2bs ? (__brev(v) >> (32 - log2bs)) : 0u;
      best = ((unsigned long long)__float_as_uint(bestd) << 32) | (0xffffffffu - ((br << 20) | (unsigned)bestk));
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
      unsigned long long o = __shfl_xor(best, s, 64);
      best = o > best ? o : best;
    }
    if (lane == 0) wave_best[wave] = best;
    __syncthreads();
    unsigned long long b2 = wave_best[lane & 15];
#pragma unroll
    for (int s = 8; s > 0; s >>= 1) {
      unsigned long long o = __shfl_xor(b2, s, 64);
      b2 = o > b2 ? o : b2;
    }
    const int sel = (int)((0xffffffffu - (unsigned)(b2 & 0xffffffffULL)) & 0xfffffu);
    if (tid == 0) {
      out[j] = sel;
      sel_xyz[0] = xyz[sel * 3 + 0]; sel_xyz[1] = xyz[sel * 3 + 1]; sel_xyz[2] = xyz[sel * 3 + 2];
    }
    __syncthreads();
  }
}

CRB_KNOB g_fps_variant = 2;      // 2 = fps2_kernel for the in-register sizes (default), 1 = fps_kernel (measurement library: A/B)
template <int PPT, bool REGS>
void launch_fps(int B, int n, int m, int log2bs, const float* xyz, int* out, hipStream_t st) {
  if (REGS && g_fps_variant == 2) hipLaunchKernelGGL((fps2_kernel<PPT>), dim3(B), dim3(FPS_THREADS), 0, st, n, m, log2bs, xyz, out);
  else hipLaunchKernelGGL((fps_kernel<PPT, REGS>), dim3(B), dim3(FPS_THREADS), 0, st, n, m, log2bs, xyz, out);
}

}  // namespace

#ifdef CRB_MEASURE
extern "C" int crb_fps_set_variant(int v) { g_fps_variant = v == 1 ? 1 : 2; return CRB_OK; }
#endif

extern "C" int crb_ball_query_stack(int B, int64_t M, float radius, int nsample, const float* new_xyz,
                                    const int32_t* new_xyz_batch_cnt, const float* xyz, const int32_t* xyz_batch_cnt,
                                    int32_t* idx, void* stream) {
  if (B <= 0 || M < 0 || nsample <= 0) return CRB_ERR_ARG;
  if (M == 0) return CRB_OK;
  hipLaunchKernelGGL(ball_query_kernel, dim3(crb_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, B, (int)M, radius,
                     nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_ball_query2_stack(int B, int64_t M, float radius_a, int nsample_a, float radius_b, int nsample_b,
                                     const float* new_xyz, const int32_t* new_xyz_batch_cnt, const float* xyz,
                                     const int32_t* xyz_batch_cnt, int32_t* idx_a, int32_t* idx_b, uint8_t* empty_a,
                                     uint8_t* empty_b, void* stream) {
  if (B <= 0 || M < 0 || nsample_a <= 0 || nsample_b <= 0 || M >= (1LL << 31)) return CRB_ERR_ARG;
  if (M == 0) return CRB_OK;
  if (radius_a <= radius_b) {                                // hits of a are hits of b: the multi-query kernel's shortcut
    constexpr int Q = 8;
    hipLaunchKernelGGL(ball_query2_multi_kernel<Q>, dim3(crb_cdiv(M, 4 * Q)), dim3(256), 0, (hipStream_t)stream, B, (int)M,
                       radius_a, nsample_a, radius_b, nsample_b, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx_a,
                       idx_b, empty_a, empty_b);
  } else {
    hipLaunchKernelGGL(ball_query2_kernel, dim3(crb_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, B, (int)M, radius_a,
                       nsample_a, radius_b, nsample_b, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx_a, idx_b,
                       empty_a, empty_b);
  }
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

// workspace of crb_ball_query2_grid_stack for n_total source points: bucket counts / starts, cursors, bucket of every point, the
// sorted point list, scan tile sums
static inline int64_t bqg_buckets(int64_t n_total) {
  int64_t nb = 1024;
  while (nb < 2 * n_total && nb < (1LL << 24)) nb <<= 1;
  return nb;
}
extern "C" int64_t crb_ball_query2_grid_workspace_bytes(int64_t n_total) {
  if (n_total <= 0) return 256;
  const int64_t nb = bqg_buckets(n_total);
  return 4 * (crb_align_up(nb + 1, 64) + crb_align_up(nb, 64) + 2 * crb_align_up(n_total, 64) + crb_align_up(crb_scan_num_tiles(nb + 1), 64) + 2 * 4096 + 128) + 1024;
}

struct BqgCount { const int* c; int64_t nb; __device__ int operator()(int64_t i) const { return i < nb ? c[i] : 0; } };
struct BqgWrite { int* start; int* cursor; int64_t nb; __device__ void operator()(int64_t i, int ex, int) const { start[i] = ex; if (i < nb) cursor[i] = ex; } };

extern "C" int crb_ball_query2_grid_stack(int B, int64_t M, float radius_a, int nsample_a, float radius_b, int nsample_b,
                                          const float* new_xyz, const int32_t* new_xyz_batch_cnt, const float* xyz,
                                          const int32_t* xyz_batch_cnt, int64_t n_total, int32_t* idx_a, int32_t* idx_b,
                                          uint8_t* empty_a, uint8_t* empty_b, void* workspace, int64_t workspace_bytes, void* stream) {
  if (B <= 0 || M < 0 || nsample_a <= 0 || nsample_b <= 0 || M >= (1LL << 31) || n_total < 0 || n_total >= (1LL << 30)) return CRB_ERR_ARG;
  if (!(radius_a <= radius_b) || !(radius_b > 0.f) || nsample_a > 64 || nsample_b > 64) return CRB_ERR_UNSUPPORTED;
  if (M == 0) return CRB_OK;
  if (workspace_bytes < crb_ball_query2_grid_workspace_bytes(n_total) || !workspace) return CRB_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nb = bqg_buckets(n_total);
  int* count = (int*)workspace;                                   // nb + 1 (the scan's input; starts after it: in place is not possible)
  int* startv = count;                                            // (exclusive scan written over the counts: the apply pass reads a tile before it writes it)
  int* cursor = count + crb_align_up(nb + 1, 64);
  int* bucket_of = cursor + crb_align_up(nb, 64);
  int* sorted = bucket_of + crb_align_up(n_total, 64);
  int* tiles = sorted + crb_align_up(n_total, 64);
  int* xyz_pref = tiles + crb_align_up(crb_scan_num_tiles(nb + 1), 64);      // B + 1 each (B <= 4095)
  int* new_pref = xyz_pref + 4096 + 64;
  if (B >= 4096) return CRB_ERR_ARG;
  BqGrid g;
  g.inv_cell = 1.0f / (radius_b * 1.001f);
  g.mask = (unsigned)(nb - 1);
  CRB_HIP(hipMemsetAsync(count, 0, (size_t)(nb + 1) * sizeof(int), st));
  hipLaunchKernelGGL(bqg_prefix_kernel, dim3(1), dim3(64), 0, st, B, xyz_batch_cnt, new_xyz_batch_cnt, xyz_pref, new_pref);
  if (n_total > 0)
    hipLaunchKernelGGL(bqg_hist_kernel, dim3(crb_cdiv(n_total, 256)), dim3(256), 0, st, g, B, (int)n_total, xyz, xyz_pref, count, bucket_of);
  const int rc = crb_device_excl_scan(BqgCount{count, nb}, BqgWrite{startv, cursor, nb}, nb + 1, tiles, nullptr, st);
  if (rc != CRB_OK) return rc;
  if (n_total > 0)
    hipLaunchKernelGGL(bqg_fill_kernel, dim3(crb_cdiv(n_total, 256)), dim3(256), 0, st, (int)n_total, bucket_of, cursor, sorted);
  hipLaunchKernelGGL(ball_query2_grid_kernel, dim3(crb_cdiv(M, 4)), dim3(256), 0, st, g, B, (int)M, radius_a, nsample_a, radius_b,
                     nsample_b, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, xyz_pref, new_pref, startv, sorted, idx_a, idx_b, empty_a, empty_b);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_ball_query2_grouped_stack(int B, int64_t M, int group, float radius_a, int nsample_a, float radius_b,
                                             int nsample_b, const float* new_xyz, const int32_t* new_xyz_batch_cnt,
                                             const float* xyz, const int32_t* xyz_batch_cnt, int32_t* idx_a, int32_t* idx_b,
                                             uint8_t* empty_a, uint8_t* empty_b, void* stream) {
  if (B <= 0 || M < 0 || nsample_a <= 0 || nsample_b <= 0 || M >= (1LL << 31)) return CRB_ERR_ARG;
  if (group <= 0 || group > 1024 || M % group) return CRB_ERR_ARG;
  if (M == 0) return CRB_OK;
  hipLaunchKernelGGL(ball_query2_grouped_kernel, dim3((unsigned)(M / group)), dim3(256), 0, (hipStream_t)stream, B, (int)M,
                     group, radius_a, nsample_a, radius_b, nsample_b, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx_a,
                     idx_b, empty_a, empty_b);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_group_points_stack(int B, int64_t M, int C, int nsample, const float* features,
                                      const int32_t* features_batch_cnt, const int32_t* idx,
                                      const int32_t* idx_batch_cnt, float* out, void* stream) {
  if (B <= 0 || M < 0 || C <= 0 || nsample <= 0) return CRB_ERR_ARG;
  if (M == 0) return CRB_OK;
  size_t lds = sizeof(float) * (size_t)nsample * (C + 1);
  if (lds > 64 * 1024) return CRB_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(group_points_kernel, dim3((unsigned)M), dim3(256), lds, (hipStream_t)stream, B, (int)M, C, nsample,
                     features, features_batch_cnt, idx, idx_batch_cnt, out);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_group_points_grad_stack(int B, int64_t M, int C, int nsample, const float* grad_out,
                                           const int32_t* idx, const int32_t* idx_batch_cnt,
                                           const int32_t* features_batch_cnt, float* grad_features /* pre-zeroed */,
                                           void* stream) {
  if (B <= 0 || M < 0 || C <= 0 || nsample <= 0) return CRB_ERR_ARG;
  if (M == 0) return CRB_OK;
  size_t lds = sizeof(float) * (size_t)nsample * (C + 1);
  if (lds > 64 * 1024) return CRB_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(group_points_grad_kernel, dim3((unsigned)M), dim3(256), lds, (hipStream_t)stream, B, (int)M, C,
                     nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_query_group_stack(int B, int64_t M, int C, int nsample, const float* xyz,
                                     const int32_t* xyz_batch_cnt, const float* features, const float* new_xyz,
                                     const int32_t* new_xyz_batch_cnt, const int32_t* idx, const uint8_t* empty_mask,
                                     float* out, void* stream) {
  if (B <= 0 || M < 0 || C < 0 || nsample <= 0) return CRB_ERR_ARG;
  if (M == 0) return CRB_OK;
  const size_t lds = sizeof(float) * 64 * (C + 4);
  if (lds > 64 * 1024) return CRB_ERR_UNSUPPORTED;
  const int64_t MP = M * nsample;
  hipLaunchKernelGGL(query_group_kernel, dim3(crb_cdiv(MP, 64)), dim3(256), lds, (hipStream_t)stream, B, MP, C, nsample,
                     xyz, xyz_batch_cnt, features, new_xyz, new_xyz_batch_cnt, idx, empty_mask, out);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_query_group_grad_stack(int B, int64_t M, int C, int nsample, const int32_t* xyz_batch_cnt,
                                          const int32_t* new_xyz_batch_cnt, const int32_t* idx,
                                          const uint8_t* empty_mask, const float* grad_out,
                                          float* grad_features /* pre-zeroed */, void* stream) {
  if (B <= 0 || M < 0 || C <= 0 || nsample <= 0) return CRB_ERR_ARG;
  if (M == 0) return CRB_OK;
  const size_t lds = sizeof(float) * 64 * (C + 1);
  if (lds > 64 * 1024) return CRB_ERR_UNSUPPORTED;
  const int64_t MP = M * nsample;
  hipLaunchKernelGGL(query_group_grad_kernel, dim3(crb_cdiv(MP, 64)), dim3(256), lds, (hipStream_t)stream, B, MP, C,
                     nsample, xyz_batch_cnt, new_xyz_batch_cnt, idx, empty_mask, grad_out, grad_features);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_query_group_rows_stack(int B, int64_t M, int C, int nsample, const float* xyz,
                                          const int32_t* xyz_batch_cnt, const float* features, const float* new_xyz,
                                          const int32_t* new_xyz_batch_cnt, const int32_t* idx,
                                          const uint8_t* empty_mask, float* out, void* stream) {
  if (B <= 0 || M < 0 || C <= 0 || nsample <= 0) return CRB_ERR_ARG;
  if (M == 0) return CRB_OK;
  const int64_t MP = M * nsample;
  hipLaunchKernelGGL(query_group_rows_kernel, dim3(crb_cdiv(MP, 64)), dim3(256), 0, (hipStream_t)stream, B, MP, C, nsample,
                     xyz, xyz_batch_cnt, features, new_xyz, new_xyz_batch_cnt, idx, empty_mask, out);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_query_group_rows_grad_stack(int B, int64_t M, int C, int nsample, const int32_t* xyz_batch_cnt,
                                               const int32_t* new_xyz_batch_cnt, const int32_t* idx,
                                               const uint8_t* empty_mask, const float* grad_out,
                                               float* grad_features /* pre-zeroed */, void* stream) {
  if (B <= 0 || M < 0 || C <= 0 || nsample <= 0) return CRB_ERR_ARG;
  if (M == 0) return CRB_OK;
  const size_t lds = sizeof(float) * 64 * C;
  if (lds > 64 * 1024) return CRB_ERR_UNSUPPORTED;
  const int64_t MP = M * nsample;
  hipLaunchKernelGGL(query_group_rows_grad_kernel, dim3(crb_cdiv(MP, 64)), dim3(256), lds, (hipStream_t)stream, B, MP, C,
                     nsample, xyz_batch_cnt, new_xyz_batch_cnt, idx, empty_mask, grad_out, grad_features);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

static int group_affine_rows_launch(int B, int64_t M, int H, int nsample, const float* xyz, const int32_t* xyz_batch_cnt,
                                    const float* P, const float* new_xyz, const int32_t* new_xyz_batch_cnt,
                                    const int32_t* idx, const uint8_t* empty_mask, const float* W1x, float* out, float* rel,
                                    float* stat, void* stream) {
  if (B <= 0 || M < 0 || H <= 0 || nsample <= 0) return CRB_ERR_ARG;
  if (stat && H != 16 && H != 32 && H != 64 && H != 128) return CRB_ERR_UNSUPPORTED;
  if (!out && !stat) return CRB_ERR_ARG;                     // out == NULL: statistics only (the compile-time-H instances)
  if (M == 0) return CRB_OK;
  const int64_t MP = M * nsample;
  const dim3 grid(crb_cdiv(MP, 64));
  hipStream_t st = (hipStream_t)stream;
#define CRB_GAF(HT)                                                                                                    \
  hipLaunchKernelGGL(group_affine_rows_kernel<HT>, grid, dim3(256), 0, st, B, MP, H, nsample, xyz, xyz_batch_cnt, P,  \
                     new_xyz, new_xyz_batch_cnt, idx, empty_mask, W1x, out, rel, stat)
  if (H == 16) CRB_GAF(16);
  else if (H == 32) CRB_GAF(32);
  else if (H == 64) CRB_GAF(64);
  else if (H == 128) CRB_GAF(128);
  else CRB_GAF(0);
#undef CRB_GAF
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_group_affine_rows_stack(int B, int64_t M, int H, int nsample, const float* xyz,
                                           const int32_t* xyz_batch_cnt, const float* P, const float* new_xyz,
                                           const int32_t* new_xyz_batch_cnt, const int32_t* idx,
                                           const uint8_t* empty_mask, const float* W1x, float* out, float* rel,
                                           void* stream) {
  return group_affine_rows_launch(B, M, H, nsample, xyz, xyz_batch_cnt, P, new_xyz, new_xyz_batch_cnt, idx, empty_mask, W1x, out,
                                  rel, nullptr, stream);
}

// the same, also writing stat (crb_group_affine_rows_grad_blocks(M, nsample), 2, H): per 64-row slab the column sums of out and
// of out^2, in the layout crb_bn_relu_forward_partials reads (H in {16, 32, 64, 128})
extern "C" int crb_group_affine_rows_stats_stack(int B, int64_t M, int H, int nsample, const float* xyz,
                                                 const int32_t* xyz_batch_cnt, const float* P, const float* new_xyz,
                                                 const int32_t* new_xyz_batch_cnt, const int32_t* idx,
                                                 const uint8_t* empty_mask, const float* W1x, float* out, float* rel,
                                                 float* stat, void* stream) {
  if (!stat) return CRB_ERR_ARG;
  return group_affine_rows_launch(B, M, H, nsample, xyz, xyz_batch_cnt, P, new_xyz, new_xyz_batch_cnt, idx, empty_mask, W1x, out,
                                  rel, stat, stream);
}

extern "C" int64_t crb_group_affine_rows_grad_blocks(int64_t M, int nsample) { return crb_cdiv(M * nsample, 64); }

// the same with the BatchNorm(+ReLU) backward of the layer's output folded in: grad_z (M*nsample, H) is the gradient w.r.t.
// relu(batchnorm(y)), y (M*nsample, H) the layer's output, dbeta / dgamma the reduced BatchNorm gradients
// (crb_bn_relu_backward with dx = NULL computes them)
extern "C" int crb_group_affine_rows_grad_bn_stack(int B, int64_t M, int H, int nsample, const int32_t* xyz_batch_cnt,
                                                   const int32_t* new_xyz_batch_cnt, const int32_t* idx,
                                                   const uint8_t* empty_mask, const float* rel, const float* grad_z,
                                                   const float* y, const float* mean, const float* invstd, const float* gamma,
                                                   const float* beta, const float* dbeta, const float* dgamma,
                                                   float* grad_P /* pre-zeroed */, float* part, void* stream) {
  if (B <= 0 || M < 0 || H <= 0 || nsample <= 0 || !y || !mean || !invstd || !gamma || !beta || !dbeta || !dgamma)
    return CRB_ERR_ARG;
  if (H != 16 && H != 32 && H != 64 && H != 128) return CRB_ERR_UNSUPPORTED;
  if (M == 0) return CRB_OK;
  const int64_t MP = M * nsample;
  const dim3 grid(crb_cdiv(MP, 64));
  hipStream_t st = (hipStream_t)stream;
  const GroupBn bn{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, y, mean, invstd, gamma, beta, dbeta, dgamma, 1.0f / (float)MP};
#define CRB_GA_CASE(HH)                                                                                              \
  if (H == HH)                                                                                                       \
    hipLaunchKernelGGL((group_affine_rows_grad_kernel<HH, true>), grid, dim3(256), 0, st, B, MP, nsample, xyz_batch_cnt, \
                       new_xyz_batch_cnt, idx, empty_mask, rel, grad_z, grad_P, part, bn);
  CRB_GA_CASE(16) CRB_GA_CASE(32) CRB_GA_CASE(64) CRB_GA_CASE(128)
#undef CRB_GA_CASE
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

// the same without the forward's saved tensors (sa_mlp_train.hip pass D): y and rel are formed again from xyz / new_xyz / P / W1x
static int group_affine_rows_grad_bn_recompute(int B, int64_t M, int H, int nsample, const float* xyz,
                                                             const int32_t* xyz_batch_cnt, const float* P, const float* new_xyz,
                                                             const int32_t* new_xyz_batch_cnt, const int32_t* idx,
                                                             const uint8_t* empty_mask, const float* W1x, const float* grad_z,
                                                             const float* mean, const float* invstd, const float* gamma,
                                                             const float* beta, const float* dbeta, const float* dgamma,
                                                             const int32_t* sorted_pair, const int32_t* sorted_row, int64_t n_src,
                                                             float* grad_P /* pre-zeroed */, float* part, void* stream,
                                                             int64_t* grad_P_fixed, float fixed_scale) {
  if (B <= 0 || M < 0 || H <= 0 || nsample <= 0 || !xyz || !P || !new_xyz || !W1x || !mean || !invstd || !gamma || !beta ||
      !dbeta || !dgamma)
    return CRB_ERR_ARG;
  if ((sorted_pair == nullptr) != (sorted_row == nullptr) || (sorted_pair && (n_src <= 0 || n_src >= (1LL << 31)))) return CRB_ERR_ARG;
  if (H != 16 && H != 32 && H != 64 && H != 128) return CRB_ERR_UNSUPPORTED;
  if (M == 0) return CRB_OK;
  const int64_t MP = M * nsample;
  const dim3 grid(crb_cdiv(MP, 64));
  hipStream_t st = (hipStream_t)stream;
  const GroupBn bn{xyz, new_xyz, P, W1x, sorted_pair, sorted_row, (int)n_src, nullptr, mean, invstd, gamma, beta, dbeta, dgamma,
                   1.0f / (float)MP, (unsigned long long*)grad_P_fixed, fixed_scale};
#define CRB_GA_CASE(HH)                                                                                                    \
  if (H == HH)                                                                                                             \
    hipLaunchKernelGGL((group_affine_rows_grad_kernel<HH, true, true>), grid, dim3(256), 0, st, B, MP, nsample, xyz_batch_cnt, \
                       new_xyz_batch_cnt, idx, empty_mask, (const float*)nullptr, grad_z, grad_P, part, bn);
  CRB_GA_CASE(16) CRB_GA_CASE(32) CRB_GA_CASE(64) CRB_GA_CASE(128)
#undef CRB_GA_CASE
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_group_affine_rows_grad_bn_recompute_stack(int B, int64_t M, int H, int nsample, const float* xyz,
                                                             const int32_t* xyz_batch_cnt, const float* P, const float* new_xyz,
                                                             const int32_t* new_xyz_batch_cnt, const int32_t* idx,
                                                             const uint8_t* empty_mask, const float* W1x, const float* grad_z,
                                                             const float* mean, const float* invstd, const float* gamma,
                                                             const float* beta, const float* dbeta, const float* dgamma,
                                                             const int32_t* sorted_pair, const int32_t* sorted_row, int64_t n_src,
                                                             float* grad_P /* pre-zeroed */, float* part, void* stream) {
  if (!grad_P) return CRB_ERR_ARG;
  return group_affine_rows_grad_bn_recompute(B, M, H, nsample, xyz, xyz_batch_cnt, P, new_xyz, new_xyz_batch_cnt, idx, empty_mask, W1x,
                                             grad_z, mean, invstd, gamma, beta, dbeta, dgamma, sorted_pair, sorted_row, n_src, grad_P,
                                             part, stream, nullptr, 0.f);
}

// deterministic form: the per-source-row sums are accumulated in grad_P_fixed (n_src, H) int64, pre-zeroed, as round(value * scale)
// (integer atomics: the result does not depend on the order in which the slabs arrive); the caller converts back
// (grad_P = grad_P_fixed / scale). scale is the caller's: 2^40 / (a power of two >= max |grad_z| * max |gamma invstd|) keeps 2^-40 of
// that magnitude per addend and overflows only beyond 2^22 times it. `part` as above (it never went through atomics).
extern "C" int crb_group_affine_rows_grad_bn_recompute_stack_fixed(int B, int64_t M, int H, int nsample, const float* xyz,
                                                                   const int32_t* xyz_batch_cnt, const float* P, const float* new_xyz,
                                                                   const int32_t* new_xyz_batch_cnt, const int32_t* idx,
                                                                   const uint8_t* empty_mask, const float* W1x, const float* grad_z,
                                                                   const float* mean, const float* invstd, const float* gamma,
                                                                   const float* beta, const float* dbeta, const float* dgamma,
                                                                   const int32_t* sorted_pair, const int32_t* sorted_row, int64_t n_src,
                                                                   int64_t* grad_P_fixed /* pre-zeroed */, float scale, float* part,
                                                                   void* stream) {
  if (!grad_P_fixed || !(scale > 0.f)) return CRB_ERR_ARG;
  return group_affine_rows_grad_bn_recompute(B, M, H, nsample, xyz, xyz_batch_cnt, P, new_xyz, new_xyz_batch_cnt, idx, empty_mask, W1x,
                                             grad_z, mean, invstd, gamma, beta, dbeta, dgamma, sorted_pair, sorted_row, n_src, nullptr,
                                             part, stream, grad_P_fixed, scale);
}

// ---- pairs in source-row order (for the sorted form of the scatter above) ---------------------------------------------------------
// key[p] = source row of pair p (start of its frame + idx[p]), n_src for the pairs of an empty ball; a stable device radix sort
// (hipCUB) of (key, p) over the bits n_src needs: sorted_row / sorted_pair. Stable = pairs of one source row stay in pair order.
__global__ __launch_bounds__(256) void pair_source_keys_kernel(int B, int64_t MP, int ns, const int* __restrict__ xyz_cnt,
                                                               const int* __restrict__ new_cnt, const int* __restrict__ idx,
                                                               const unsigned char* __restrict__ empty, int n_src,
                                                               int* __restrict__ key, int* __restrict__ val) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= MP) return;
  const int m = (int)(p / ns);
  int k = n_src;
  if (!empty[m]) {
    int start;
    locate_batch(new_cnt, B, m, xyz_cnt, &start);
    k = start + idx[p];
  }
  key[p] = k;
  val[p] = (int)p;
}

static size_t pair_sort_temp_bytes(int64_t n) {
  size_t b = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, (const int*)nullptr, (int*)nullptr, (const int*)nullptr, (int*)nullptr, (int)n,
                                           0, 32, (hipStream_t)0);
  return b;
}

extern "C" int64_t crb_pair_sort_workspace_bytes(int64_t M, int nsample) {
  const int64_t n = M * nsample;
  if (n <= 0) return 256;
  return 2 * crb_align_up(4 * n, 256) + crb_align_up((int64_t)pair_sort_temp_bytes(n), 256) + 256;
}

extern "C" int crb_pair_sort_by_source(int B, int64_t M, int nsample, const int32_t* xyz_batch_cnt, const int32_t* new_xyz_batch_cnt,
                                       const int32_t* idx, const uint8_t* empty_mask, int64_t n_src, int32_t* sorted_pair,
                                       int32_t* sorted_row, void* workspace, int64_t workspace_bytes, void* stream) {
  const int64_t n = M * nsample;
  if (B <= 0 || M < 0 || nsample <= 0 || n >= (1LL << 31) || n_src <= 0 || n_src >= (1LL << 31) - 1 || !sorted_pair || !sorted_row)
    return CRB_ERR_ARG;
  if (n == 0) return CRB_OK;
  if (!workspace || workspace_bytes < crb_pair_sort_workspace_bytes(M, nsample)) return CRB_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  CrbArena ar(workspace, (size_t)workspace_bytes);
  int* kin = ar.take<int>(n);
  int* vin = ar.take<int>(n);
  const size_t tb = pair_sort_temp_bytes(n);
  char* temp = ar.take<char>((int64_t)tb);
  if (!ar.ok) return CRB_ERR_WORKSPACE;
  hipLaunchKernelGGL(pair_source_keys_kernel, dim3(crb_cdiv(n, 256)), dim3(256), 0, st, B, n, nsample, xyz_batch_cnt,
                     new_xyz_batch_cnt, idx, empty_mask, (int)n_src, kin, vin);
  int bits = 1;
  while ((1LL << bits) <= n_src) ++bits;                     // keys 0 .. n_src
  size_t tbytes = tb;
  if (hipcub::DeviceRadixSort::SortPairs(temp, tbytes, (const int*)kin, sorted_row, (const int*)vin, sorted_pair, (int)n, 0, bits, st) !=
      hipSuccess)
    return CRB_ERR_LAUNCH;
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

// ------------------------------------------------------------------------------------------------ rows per frame
// The stacked layout keeps the rows of a frame together, frames in order (that is what xyz_batch_cnt means to every kernel of this
// file), so the frame-index column is non-decreasing and the rows of frame b are [lower_bound(b), lower_bound(b + 1)). One wave per
// frame, 64-ary search: 64 probes per round, 4 rounds for 320 k rows. (The module-level form of the reference,
// voxel_set_abstraction.py:321-323 `(xyz_bs_idxs == bs_idx).sum()` per frame, is B reductions over the column; a scatter_add of
// ones is B hot atomics: 46 us for 320 k rows.)
template <typename T>
__device__ int sorted_lower_bound_wave(const T* __restrict__ key, int64_t stride, int n, int target) {
  const int lane = threadIdx.x & 63;
  int lo = 0, hi = n;                                        // the answer is in [lo, hi]
  while (hi > lo) {
    const int step = (hi - lo + 63) / 64;
    const int64_t i = (int64_t)lo + (int64_t)lane * step;
    const bool ge = i >= hi || (int)key[i * stride] >= target;
    const unsigned long long m = __ballot(ge);
    if (m & 1ULL) return lo;                                 // key[lo] >= target
    const int f = m ? __ffsll((long long)m) - 1 : 64;       // lane f - 1 is below the target, lane f is not (or past hi)
    const int64_t nhi = (int64_t)lo + (int64_t)f * step;
    lo = lo + (f - 1) * step + 1;
    hi = (f == 64 || nhi > hi) ? hi : (int)nhi;
  }
  return lo;
}

template <typename T>
__global__ __launch_bounds__(64) void sorted_key_counts_kernel(const T* __restrict__ key, int64_t stride, int n, int* __restrict__ counts) {
  const int b = blockIdx.x;
  const int lo = sorted_lower_bound_wave(key, stride, n, b), hi = sorted_lower_bound_wave(key, stride, n, b + 1);
  if (threadIdx.x == 0) counts[b] = hi - lo;
}

extern "C" int crb_sorted_key_counts(const void* key, int key_is_float, int64_t stride, int64_t n, int B, int32_t* counts, void* stream) {
  if (B <= 0 || n < 0 || n >= (1LL << 31) || stride <= 0 || !counts || (n > 0 && !key)) return CRB_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (key_is_float)
    hipLaunchKernelGGL(sorted_key_counts_kernel<float>, dim3(B), dim3(64), 0, st, (const float*)key, stride, (int)n, counts);
  else
    hipLaunchKernelGGL(sorted_key_counts_kernel<int>, dim3(B), dim3(64), 0, st, (const int*)key, stride, (int)n, counts);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_group_affine_rows_grad_stack(int B, int64_t M, int H, int nsample, const int32_t* xyz_batch_cnt,
                                                const int32_t* new_xyz_batch_cnt, const int32_t* idx,
                                                const uint8_t* empty_mask, const float* rel, const float* grad_out,
                                                float* grad_P /* pre-zeroed */, float* part, void* stream) {
  if (B <= 0 || M < 0 || H <= 0 || nsample <= 0) return CRB_ERR_ARG;
  if (H != 16 && H != 32 && H != 64 && H != 128) return CRB_ERR_UNSUPPORTED;
  if (M == 0) return CRB_OK;
  const int64_t MP = M * nsample;
  const dim3 grid(crb_cdiv(MP, 64));
  hipStream_t st = (hipStream_t)stream;
#define CRB_GA_CASE(HH)                                                                                              \
  if (H == HH)                                                                                                       \
    hipLaunchKernelGGL((group_affine_rows_grad_kernel<HH, false>), grid, dim3(256), 0, st, B, MP, nsample, xyz_batch_cnt, \
                       new_xyz_batch_cnt, idx, empty_mask, rel, grad_out, grad_P, part, GroupBn{});
  CRB_GA_CASE(16) CRB_GA_CASE(32) CRB_GA_CASE(64) CRB_GA_CASE(128)
#undef CRB_GA_CASE
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_farthest_point_sample(int B, int n, int m, const float* xyz, float* temp, int32_t* out_idx,
                                         void* stream) {
  if (B <= 0 || n <= 0 || m < 0) return CRB_ERR_ARG;
  if (m == 0) return CRB_OK;
  if (n >= (1 << 20)) return CRB_ERR_UNSUPPORTED;
  if (n > FPS_THREADS * FPS_REG_MAX_PER_THREAD && !temp) return CRB_ERR_WORKSPACE;
  int log2bs = 0;
  while ((2 << log2bs) <= n && log2bs < 10) ++log2bs;       // bs = min(2^floor(log2 n), 1024)
  hipStream_t st = (hipStream_t)stream;
  const int ppt = (n + FPS_THREADS - 1) / FPS_THREADS;
  if (ppt <= 4) launch_fps<4, true>(B, n, m, log2bs, xyz, out_idx, st);
  else if (ppt <= 8) launch_fps<8, true>(B, n, m, log2bs, xyz, out_idx, st);
  else if (ppt <= 20) launch_fps<20, true>(B, n, m, log2bs, xyz, out_idx, st);
  else if (ppt <= 40) launch_fps<40, false>(B, n, m, log2bs, xyz, out_idx, st);
  else hipLaunchKernelGGL(fps_large_kernel, dim3(B), dim3(FPS_THREADS), 0, st, n, m, log2bs, xyz, temp, out_idx);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_three_nn_stack(int B, int64_t N, const float* unknown, const int32_t* unknown_batch_cnt,
                                  const float* known, const int32_t* known_batch_cnt, float* dist2, int32_t* idx,
                                  void* stream) {
  if (B <= 0 || N < 0) return CRB_ERR_ARG;
  if (N == 0) return CRB_OK;
  hipLaunchKernelGGL(three_nn_kernel, dim3(crb_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, B, (int)N, unknown,
                     unknown_batch_cnt, known, known_batch_cnt, dist2, idx);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_three_interpolate_stack(int64_t N, int C, const float* features, const int32_t* idx,
                                           const float* weight, float* out, void* stream) {
  if (N < 0 || C <= 0) return CRB_ERR_ARG;
  if (N == 0) return CRB_OK;
  hipLaunchKernelGGL(three_interp_kernel, dim3(crb_cdiv(N * C, 256)), dim3(256), 0, (hipStream_t)stream, (int)N, C,
                     features, idx, weight, out);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_three_interpolate_grad_stack(int64_t N, int C, const float* grad_out, const int32_t* idx,
                                                const float* weight, float* grad_features /* pre-zeroed */,
                                                void* stream) {
  if (N < 0 || C <= 0) return CRB_ERR_ARG;
  if (N == 0) return CRB_OK;
  hipLaunchKernelGGL(three_interp_grad_kernel, dim3(crb_cdiv(N * C, 256)), dim3(256), 0, (hipStream_t)stream, (int)N, C,
                     grad_out, idx, weight, grad_features);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

// ---- voxel centres (common_utils.get_voxel_centers, pcdet/utils/common_utils.py:63-80): coords (n, 3) [z, y, x] integers with a row
//      stride (a column slice of the (n, 4) [b, z, y, x] index tensor) -> (n, 3) xyz = (coord + 0.5) * (voxel_size * downsample) + range
//      minimum, the torch expression's operations in its order. flip + cast + four elementwise launches per VSA level as one.
namespace {
__global__ __launch_bounds__(256) void voxel_centers_kernel(const int32_t* __restrict__ coords, int64_t row_stride, int64_t n, float sx,
                                                            float sy, float sz, float mx, float my, float mz, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t* c = coords + i * row_stride;
  out[i * 3 + 0] = ((float)c[2] + 0.5f) * sx + mx;
  out[i * 3 + 1] = ((float)c[1] + 0.5f) * sy + my;
  out[i * 3 + 2] = ((float)c[0] + 0.5f) * sz + mz;
}
}  // namespace

extern "C" int crb_voxel_centers(const int32_t* coords_zyx, int64_t row_stride, int64_t n, const float* scaled_voxel_size,
                                 const float* range_min, float* out, void* stream) {
  if (n < 0 || row_stride < 3 || !scaled_voxel_size || !range_min) return CRB_ERR_ARG;
  if (n == 0) return CRB_OK;
  if (!coords_zyx || !out) return CRB_ERR_ARG;
  hipLaunchKernelGGL(voxel_centers_kernel, dim3(crb_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, coords_zyx, row_stride, n,
                     scaled_voxel_size[0], scaled_voxel_size[1], scaled_voxel_size[2], range_min[0], range_min[1], range_min[2], out);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}
