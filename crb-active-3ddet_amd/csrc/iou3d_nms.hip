// Rotated BEV overlap / IoU, 3-D IoU and NMS for gfx950 (rows a13, a14 of SURVEY §8).
//
// Replaces pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu (boxes_overlap_kernel :236, boxes_iou_bev_kernel :251,
// nms_kernel :267, nms_normal_kernel :328) and the host half of nms_gpu in iou3d_nms.cpp:90-136 (cudaMalloc of the
// mask, D2H copy, serial CPU greedy scan, cudaFree).
//
// Parity contract: NMS picks must be identical to the reference, so the rotated-rectangle overlap keeps the
// reference's geometric definition step for step (corner rotation, 4x4 edge intersections with its cross-product sign
// test, corner-in-box test with the 1e-2 margin, ordering of the hull points by atan2 about their centroid, shoelace
// fan) and the same f32 operation order; -ffp-contract=off keeps hipcc from fusing a*b+c differently from gcc.
//
// MI355X design:
//  * suppression mask: one 64-lane wave per 64x64 tile, the tile's 64 column boxes staged in LDS, one 64-bit word per
//    row written with a single store; tiles below the diagonal are never needed by the greedy scan and are skipped
//    (the reference computes them);
//  * greedy scan on the device (one wave per frame): the wave keeps the whole `removed` bitmap in registers
//    (lane l owns words l, l+64, ...), jumps from kept box to kept box with ffs on the current word, and ORs in only the
//    mask rows of kept boxes — no mask D2H, no host loop, launch-count independent of the number of boxes;
//  * batched over frames in one launch (grid.z / grid.x = frame) with per-frame box counts read on the device.
#include "crb_common.h"
#include "../../include/crb_hip.h"

namespace {

constexpr float kEps = 1e-8f;

struct P2 { float x, y; };

__device__ __forceinline__ float cross3(P2 p1, P2 p2, P2 p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

__device__ __forceinline__ bool bbox_cross(P2 p1, P2 p2, P2 q1, P2 q2) {
  return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
         fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

// segment (p0,p1) x segment (q0,q1); returns true and the crossing point when they properly cross
__device__ __forceinline__ bool seg_cross(P2 p1, P2 p0, P2 q1, P2 q0, P2& out) {
  if (!bbox_cross(p0, p1, q0, q1)) return false;
  float s1 = cross3(q0, p1, p0);
  float s2 = cross3(p1, q1, p0);
  float s3 = cross3(p0, q1, q0);
  float s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
  float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > kEps) {
    out.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    out.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    out.x = (b0 * c1 - b1 * c0) / D;
    out.y = (a1 * c0 - a0 * c1) / D;
  }
  return true;
}

__device__ __forceinline__ bool in_box_margin(const float* b, P2 p) {
  const float margin = 1e-2f;
  float ac = cosf(-b[6]), as = sinf(-b[6]);
  float rx = (p.x - b[0]) * ac + (p.y - b[1]) * (-as);
  float ry = (p.x - b[0]) * as + (p.y - b[1]) * ac;
  return fabsf(rx) < b[3] / 2 + margin && fabsf(ry) < b[4] / 2 + margin;
}

__device__ __forceinline__ void box_corners(const float* b, P2* c /*5*/) {
  float hx = b[3] / 2, hy = b[4] / 2;
  float x1 = b[0] - hx, y1 = b[1] - hy, x2 = b[0] + hx, y2 = b[1] + hy;
  float ca = cosf(b[6]), sa = sinf(b[6]);
  const float px[4] = {x1, x2, x2, x1};
  const float py[4] = {y1, y1, y2, y2};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float nx = (px[k] - b[0]) * ca + (py[k] - b[1]) * (-sa) + b[0];
    float ny = (px[k] - b[0]) * sa + (py[k] - b[1]) * ca + b[1];
    c[k].x = nx; c[k].y = ny;
  }
  c[4] = c[0];
}

// rotated-rectangle intersection area (reference: box_overlap, iou3d_nms_kernel.cu:104-225)
__device__ float rect_overlap(const float* a, const float* b) {
  P2 ca[5], cb[5];
  box_corners(a, ca);
  box_corners(b, cb);
  P2 pts[16];
  float sx = 0.f, sy = 0.f;
  int cnt = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      P2 q;
      if (seg_cross(ca[i + 1], ca[i], cb[j + 1], cb[j], q)) {
        sx = sx + q.x; sy = sy + q.y;
        pts[cnt++] = q;
      }
    }
  for (int k = 0; k < 4; ++k) {
    if (in_box_margin(a, cb[k])) { sx = sx + cb[k].x; sy = sy + cb[k].y; pts[cnt++] = cb[k]; }
    if (in_box_margin(b, ca[k])) { sx = sx + ca[k].x; sy = sy + ca[k].y; pts[cnt++] = ca[k]; }
  }
  float cx = sx / cnt, cy = sy / cnt;    // cnt == 0 -> NaN centre, loops below do nothing, area 0 (as the reference)
  // bubble sort by polar angle about the centroid (same comparison sequence as the reference => same permutation)
  float ang[16];
  for (int k = 0; k < cnt; ++k) ang[k] = atan2f(pts[k].y - cy, pts[k].x - cx);
  for (int j = 0; j < cnt - 1; ++j)
    for (int i = 0; i < cnt - j - 1; ++i)
      if (ang[i] > ang[i + 1]) {
        P2 t = pts[i]; pts[i] = pts[i + 1]; pts[i + 1] = t;
        float ta = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = ta;
      }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; ++k) {
    float ax = pts[k].x - pts[0].x, ay = pts[k].y - pts[0].y;
    float bx = pts[k + 1].x - pts[0].x, by = pts[k + 1].y - pts[0].y;
    area += ax * by - ay * bx;
  }
  return fabsf(area) / 2.0f;
}

__device__ __forceinline__ float iou_bev_rot(const float* a, const float* b) {
  float sa = a[3] * a[4], sb = b[3] * b[4];
  float so = rect_overlap(a, b);
  return so / fmaxf(sa + sb - so, kEps);
}

__device__ __forceinline__ float iou_bev_normal(const float* a, const float* b) {
  float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
  float inter = w * h;
  return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, kEps);
}

// mode 0: overlap area, 1: BEV IoU, 2: 3-D IoU (iou3d_nms_utils.py:48-81 fused)
template <int MODE>
__global__ __launch_bounds__(256) void pairwise_kernel(const float* __restrict__ A, int na, const float* __restrict__ Bx,
                                                       int nb, float* __restrict__ out) {
  __shared__ float sb[64 * 7];
  const int b0 = blockIdx.x * 64, a0 = blockIdx.y * 4;
  for (int t = threadIdx.x; t < 64 * 7; t += 256) {
    int j = b0 + t / 7;
    sb[t] = j < nb ? Bx[(int64_t)b0 * 7 + t] : 0.f;
  }
  __syncthreads();
  const int ai = a0 + (threadIdx.x >> 6), bj = b0 + (threadIdx.x & 63);
  if (ai >= na || bj >= nb) return;
  float a[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) a[k] = A[(int64_t)ai * 7 + k];
  const float* b = sb + (threadIdx.x & 63) * 7;
  float r;
  if (MODE == 0) r = rect_overlap(a, b);
  else if (MODE == 1) r = iou_bev_rot(a, b);
  else {
    float ov = rect_overlap(a, b);
    float amax = a[2] + a[5] / 2, amin = a[2] - a[5] / 2, bmax = b[2] + b[5] / 2, bmin = b[2] - b[5] / 2;
    float oh = fmaxf(fminf(amax, bmax) - fmaxf(amin, bmin), 0.f);
    float o3 = ov * oh;
    float va = a[3] * a[4] * a[5], vb = b[3] * b[4] * b[5];
    r = o3 / fmaxf(va + vb - o3, 1e-6f);
  }
  out[(int64_t)ai * nb + bj] = r;
}

// suppression mask. grid (col_blocks, col_blocks, B), one wave per tile. boxes (B, nmax, 7) score-sorted.
template <bool ROTATED>
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, const int* __restrict__ counts,
                                                      int nmax, float thresh, unsigned long long* __restrict__ mask,
                                                      int col_blocks, int nlimit, const int* __restrict__ done) {
  const int row_blk = blockIdx.y, col_blk = blockIdx.x, f = blockIdx.z;
  if (row_blk > col_blk) return;                       // never read by the greedy scan
  if (done && done[f]) return;                         // the prefix stage already holds this frame's result
  const int n = min(counts ? min(counts[f], nmax) : nmax, nlimit);
  if (row_blk * 64 >= n || col_blk * 64 >= n) return;
  const float* fb = boxes + (int64_t)f * nmax * 7;
  __shared__ float cb[64 * 7];
  const int col_size = min(n - col_blk * 64, 64);
  for (int t = threadIdx.x; t < 64 * 7; t += 64) cb[t] = (t / 7 < col_size) ? fb[(int64_t)col_blk * 64 * 7 + t] : 0.f;
  __syncthreads();
  const int i = row_blk * 64 + threadIdx.x;
  if (i >= n) return;
  float a[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) a[k] = fb[(int64_t)i * 7 + k];
  unsigned long long t = 0;
  const int start = (row_blk == col_blk) ? threadIdx.x + 1 : 0;
  if constexpr (ROTATED) {
    // Two passes. (1) cheap reject with circumscribed circles: boxes whose BEV centres are farther apart than the sum of
    // their half-diagonals (+1e-4 relative slack) cannot intersect, the reference's polygon clipping finds no vertex for
    // them and returns overlap 0 -> IoU 0 -> never above a positive threshold. (2) the rotated-overlap arithmetic only for
    // the survivors: a wave runs max-over-lanes(survivors) iterations instead of 64 (proposals are score-sorted, i.e.
    // spatially random: typically a handful of the 64 columns are near a given row box).
    const float ra = 0.5f * sqrtf(a[3] * a[3] + a[4] * a[4]);
    unsigned long long cand = 0;
    for (int j = start; j < col_size; ++j) {
      const float* b = cb + j * 7;
      const float rb = 0.5f * sqrtf(b[3] * b[3] + b[4] * b[4]);
      const float dx = a[0] - b[0], dy = a[1] - b[1], rs = ra + rb;
      if (dx * dx + dy * dy <= rs * rs * 1.0001f + 1e-6f) cand |= 1ULL << j;
    }
    if (!(thresh > 0.f)) cand = (col_size >= 64 ? ~0ULL : ((1ULL << col_size) - 1ULL)) & (~0ULL << start);  // degenerate threshold: test everything
    while (cand) {
      const int j = __ffsll((long long)cand) - 1;
      cand &= cand - 1;
      if (iou_bev_rot(a, cb + j * 7) > thresh) t |= 1ULL << j;
    }
  } else {
    for (int j = start; j < col_size; ++j)
      if (iou_bev_normal(a, cb + j * 7) > thresh) t |= 1ULL << j;
  }
  mask[((int64_t)f * nmax + i) * col_blocks + col_blk] = t;
}

// greedy scan: one wave per frame. keep (B, max_keep) int32 (indices into the sorted boxes, -1 padded), num_keep (B).
constexpr int kMaxWordsPerLane = 8;   // up to 64*8*64 = 32768 boxes

__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* __restrict__ mask,
                                                      const int* __restrict__ counts, int nmax, int col_blocks,
                                                      int max_keep, int* __restrict__ keep, int* __restrict__ num_keep,
                                                      int nlimit, int* __restrict__ done, int prefix_stage) {
  const int f = blockIdx.x, lane = threadIdx.x;
  if (done && !prefix_stage && done[f]) return;
  const int n_all = counts ? min(counts[f], nmax) : nmax;
  const int n = min(n_all, nlimit);
  const unsigned long long* fm = mask + (int64_t)f * nmax * col_blocks;
  int* fk = keep + (int64_t)f * max_keep;
  unsigned long long rem[kMaxWordsPerLane];
#pragma unroll
  for (int w = 0; w < kMaxWordsPerLane; ++w) rem[w] = 0ULL;
  int nk = 0;
  const int nblk = (n + 63) >> 6;
  for (int blk = 0; blk < nblk && nk < max_keep; ++blk) {
    // current word of the removed bitmap, broadcast from its owner lane
    unsigned long long mine = 0ULL;
#pragma unroll
    for (int w = 0; w < kMaxWordsPerLane; ++w)
      if (w == (blk >> 6)) mine = rem[w];
    unsigned long long cur = __shfl(mine, blk & 63, 64);
    const int valid = min(n - blk * 64, 64);
    if (valid < 64) cur |= ~0ULL << valid;               // boxes beyond n are "removed"
    while (~cur != 0ULL && nk < max_keep) {
      const int bit = __ffsll((long long)~cur) - 1;      // next surviving box of this block
      const int i = blk * 64 + bit;
      if (lane == 0) fk[nk] = i;
      ++nk;
      cur |= 1ULL << bit;
      // OR the kept box's mask row into the bitmap (only words >= blk are ever written by the mask kernel)
      const unsigned long long* row = fm + (int64_t)i * col_blocks;
#pragma unroll
      for (int w = 0; w < kMaxWordsPerLane; ++w) {
        const int word = w * 64 + lane;
        if (word >= blk && word < nblk) rem[w] |= row[word];
      }
      unsigned long long m2 = 0ULL;
#pragma unroll
      for (int w = 0; w < kMaxWordsPerLane; ++w)
        if (w == (blk >> 6)) m2 = rem[w];
      cur |= __shfl(m2, blk & 63, 64);
    }
  }
  for (int k = nk + lane; k < max_keep; k += 64) fk[k] = -1;
  if (lane == 0) {
    num_keep[f] = nk;
    // prefix stage: the greedy pick of box i depends on the kept boxes before i only, so the first max_keep picks among the
    // first nlimit boxes ARE the answer once max_keep is reached (or the frame has no further box)
    if (prefix_stage) done[f] = (nk >= max_keep || n_all <= nlimit) ? 1 : 0;
  }
}

}  // namespace

extern "C" int crb_boxes_pairwise(const float* boxes_a, int64_t na, const float* boxes_b, int64_t nb, float* out,
                                  int mode, void* stream) {
  if (na < 0 || nb < 0 || mode < 0 || mode > 2) return CRB_ERR_ARG;
  if (na == 0 || nb == 0) return CRB_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(crb_cdiv(nb, 64), crb_cdiv(na, 4));
  if (mode == 0) hipLaunchKernelGGL(pairwise_kernel<0>, grid, dim3(256), 0, st, boxes_a, (int)na, boxes_b, (int)nb, out);
  if (mode == 1) hipLaunchKernelGGL(pairwise_kernel<1>, grid, dim3(256), 0, st, boxes_a, (int)na, boxes_b, (int)nb, out);
  if (mode == 2) hipLaunchKernelGGL(pairwise_kernel<2>, grid, dim3(256), 0, st, boxes_a, (int)na, boxes_b, (int)nb, out);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

// prefix stage of the batched NMS: with max_keep << nmax (training proposals: 512 of 9,000, threshold 0.8 — nearly every box
// survives, the scan is over after ~600 rows) only the leading block of the suppression matrix is ever read. Stage A computes
// and scans the first nms_prefix(..) boxes; frames that reached max_keep there (or have no more boxes) are final, the full
// matrix is computed only for the others (their workgroups of stage B leave at once otherwise). Same picks by construction.
static inline int nms_prefix(int64_t nmax, int max_keep) {
  int64_t p = ((int64_t)max_keep * 2 + 63) / 64 * 64;
  if (p < 1024) p = 1024;
  return nmax >= 2 * p ? (int)p : 0;                   // 0 = single stage
}

extern "C" int64_t crb_nms_workspace_bytes(int B, int64_t nmax) {
  int64_t cb = (nmax + 63) / 64;
  return crb_align_up((int64_t)B * nmax * cb * 8, 256) + crb_align_up((int64_t)B * 4, 256) + 256;
}

extern "C" int crb_nms_batched(const float* boxes_sorted, const int32_t* counts, int B, int64_t nmax, float thresh,
                               int rotated, int max_keep, int32_t* keep, int32_t* num_keep, void* workspace,
                               int64_t workspace_bytes, void* stream) {
  if (B <= 0 || nmax < 0 || max_keep <= 0) return CRB_ERR_ARG;
  if (nmax > 64 * 64 * kMaxWordsPerLane) return CRB_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (nmax == 0) {
    CRB_HIP(hipMemsetAsync(num_keep, 0, sizeof(int) * B, st));
    CRB_HIP(hipMemsetAsync(keep, 0xff, sizeof(int) * (size_t)B * max_keep, st));
    return CRB_OK;
  }
  if (workspace_bytes < crb_nms_workspace_bytes(B, nmax) - 256 || !workspace) return CRB_ERR_WORKSPACE;
  const int cb = (int)((nmax + 63) / 64);
  unsigned long long* mask = (unsigned long long*)workspace;
  int* done = reinterpret_cast<int*>(static_cast<char*>(workspace) + crb_align_up((int64_t)B * nmax * cb * 8, 256));
  const int prefix = nms_prefix(nmax, max_keep);
  if (prefix) {
    const int cbp = prefix / 64;
    dim3 gp(cbp, cbp, B);
    if (rotated)
      hipLaunchKernelGGL(nms_mask_kernel<true>, gp, dim3(64), 0, st, boxes_sorted, counts, (int)nmax, thresh, mask, cb, prefix,
                         (const int*)nullptr);
    else
      hipLaunchKernelGGL(nms_mask_kernel<false>, gp, dim3(64), 0, st, boxes_sorted, counts, (int)nmax, thresh, mask, cb, prefix,
                         (const int*)nullptr);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(64), 0, st, mask, counts, (int)nmax, cb, max_keep, keep, num_keep, prefix,
                       done, 1);
  }
  const int* done_in = prefix ? done : nullptr;
  dim3 grid(cb, cb, B);
  if (rotated)
    hipLaunchKernelGGL(nms_mask_kernel<true>, grid, dim3(64), 0, st, boxes_sorted, counts, (int)nmax, thresh, mask, cb,
                       (int)nmax, done_in);
  else
    hipLaunchKernelGGL(nms_mask_kernel<false>, grid, dim3(64), 0, st, boxes_sorted, counts, (int)nmax, thresh, mask, cb,
                       (int)nmax, done_in);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(64), 0, st, mask, counts, (int)nmax, cb, max_keep, keep, num_keep,
                     (int)nmax, prefix ? done : (int*)nullptr, 0);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}
