// points-in-boxes and RoI-aware 3-D pooling for gfx950 (row a15 of SURVEY §8 + the PartA2 pool named by north_star).
//
// Replaces pcdet/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu: points_in_boxes_kernel :313,
// generate_pts_mask_for_box3d :39, collect_inside_pts_for_box3d :78, roiaware_{max,avg}pool3d :111/:160 and their
// backward kernels :236/:261 (pybind surface roiaware_pool3d.cpp:172-177).
//
// Point-in-box test: |z-cz| <= dz/2 and the rotated |lx| < dx/2 + 1e-5, |ly| < dy/2 + 1e-5, evaluated like the
// reference's device function (f32 sin/cos of -heading, the half-extent comparisons promoted to double by its
// `/ 2.0` literals, roiaware_pool3d_kernel.cu:23-36).
//
// MI355X design: the boxes of a frame are staged once per workgroup in LDS together with their sin/cos (the reference
// recomputes sin/cos for every point x box pair); the pool's per-box point collection runs one WAVE per box over
// 64-point chunks with ballot-ranked ordered appends instead of one serial thread per box.
#include "crb_common.h"
#include "../../include/crb_hip.h"

namespace {

struct BoxLds { float cx, cy, cz, dx, dy, dz, ca, sa; };

__device__ __forceinline__ bool pt_in_box(float x, float y, float z, const BoxLds& b, float& lx, float& ly) {
  if ((double)fabsf(z - b.cz) > (double)b.dz / 2.0) return false;
  const float sx = x - b.cx, sy = y - b.cy;
  lx = sx * b.ca + sy * (-b.sa);
  ly = sx * b.sa + sy * b.ca;
  const double margin = (double)1e-5f;
  return ((double)fabsf(lx) < (double)b.dx / 2.0 + margin) && ((double)fabsf(ly) < (double)b.dy / 2.0 + margin);
}

__device__ __forceinline__ BoxLds load_box(const float* p) {
  BoxLds b;
  b.cx = p[0]; b.cy = p[1]; b.cz = p[2]; b.dx = p[3]; b.dy = p[4]; b.dz = p[5];
  b.ca = cosf(-p[6]); b.sa = sinf(-p[6]);
  return b;
}

constexpr int PIB_CHUNK = 256;

// boxes (B,T,7), pts (B,M,3) -> idx (B,M): first box containing the point or -1
__global__ __launch_bounds__(256) void points_in_boxes_kernel(int T, int M, const float* __restrict__ boxes,
                                                              const float* __restrict__ pts, int* __restrict__ out) {
  __shared__ BoxLds sb[PIB_CHUNK];
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float* fb = boxes + (int64_t)b * T * 7;
  float x = 0.f, y = 0.f, z = 0.f;
  if (i < M) {
    const float* p = pts + ((int64_t)b * M + i) * 3;
    x = p[0]; y = p[1]; z = p[2];
  }
  int found = -1;
  for (int t0 = 0; t0 < T; t0 += PIB_CHUNK) {
    __syncthreads();
    const int tc = min(PIB_CHUNK, T - t0);
    if ((int)threadIdx.x < tc) sb[threadIdx.x] = load_box(fb + (int64_t)(t0 + threadIdx.x) * 7);
    __syncthreads();
    if (found < 0 && i < M) {
      float lx, ly;
      for (int k = 0; k < tc; ++k)
        if (pt_in_box(x, y, z, sb[k], lx, ly)) { found = t0 + k; break; }
    }
  }
  if (i < M) out[(int64_t)b * M + i] = found;
}

// ---------------------------------------------------------------------------------------------- GT point statistics
// Replaces the per-frame / per-class Python loops of Detector3DTemplate.post_processing (detector3d_template.py:236-268):
// for every class, points_in_boxes_gpu over that class's gt boxes, then `(idx == i).sum()` per unique index on the host.
// One launch counts, for every point of every frame, the FIRST gt box of EACH class that contains it (the reference runs
// the first-hit kernel once per class over that class's boxes only, so a point may count for one box per class); a
// second launch (one wave per frame x class) reduces the counts to the four recorded numbers.
//
// pts rows are the stacked batch layout (N, stride) [b, x, y, z, ...] with frame offsets; gt (B,G,8) rows
// [x,y,z,dx,dy,dz,heading,label], label 0 = collate padding.
struct GtBoxLds { BoxLds box; int cls; };

__global__ __launch_bounds__(256) void gt_point_count_kernel(int G, int C, int stride, const float* __restrict__ pts,
                                                             const int* __restrict__ off, const float* __restrict__ gt,
                                                             int* __restrict__ cnt, int* __restrict__ bg) {
  __shared__ GtBoxLds sb[PIB_CHUNK];
  const int b = blockIdx.y;
  const int n0 = off[b], n1 = off[b + 1];
  const int base = n0 + (int)blockIdx.x * 256;
  if (base >= n1) return;                                   // uniform per workgroup
  const int i = base + (int)threadIdx.x;
  const bool active = i < n1;
  float x = 0.f, y = 0.f, z = 0.f;
  if (active) {
    const float* p = pts + (int64_t)i * stride;
    x = p[1]; y = p[2]; z = p[3];
  }
  const float* fb = gt + (int64_t)b * G * 8;
  uint32_t hit = 0;                                         // bit c: a box of class c already owns this point
  for (int t0 = 0; t0 < G; t0 += PIB_CHUNK) {
    __syncthreads();
    const int tc = min(PIB_CHUNK, G - t0);
    if ((int)threadIdx.x < tc) {
      const float* r = fb + (int64_t)(t0 + threadIdx.x) * 8;
      sb[threadIdx.x].box = load_box(r);
      const float lab = r[7];
      const int cls = (int)lab;
      sb[threadIdx.x].cls = ((float)cls == lab && cls >= 1 && cls <= C) ? cls - 1 : -1;
    }
    __syncthreads();
    if (active) {
      float lx, ly;
      for (int k = 0; k < tc; ++k) {
        const int cls = sb[k].cls;
        if (cls < 0 || ((hit >> cls) & 1u)) continue;
        if (pt_in_box(x, y, z, sb[k].box, lx, ly)) {
          hit |= 1u << cls;
          atomicAdd(&cnt[(int64_t)b * G + t0 + k], 1);
        }
      }
    }
  }
  // background points per class (a point owned by no box of the class), one atomic per wave and class
  for (int c = 0; c < C; ++c) {
    const unsigned long long m = __ballot(active && !((hit >> c) & 1u));
    if (crb_lane() == 0 && m) atomicAdd(&bg[b * C + c], __popcll(m));
  }
}

// one wave per (frame, class): stats[(b*C+c)*5 + ..] = {num_bbox, n_counted, mean, median, variance}
// n_counted = boxes of the class owning >= 1 point, minus the first of them when the frame has no background point for the
// class (the reference drops the first entry of torch.unique's counts, assuming it is the -1 bin: :258).
// mean = f32(sum)/n, median = lower median (torch.median), variance = population variance (unbiased=False).
__global__ __launch_bounds__(64) void gt_point_stats_kernel(int G, int C, const float* __restrict__ gt,
                                                            const int* __restrict__ cnt, const int* __restrict__ bg,
                                                            float* __restrict__ stats) {
  const int b = blockIdx.x / C, c = blockIdx.x % C;
  const int lane = crb_lane();
  const float* fb = gt + (int64_t)b * G * 8;
  const int* cb = cnt + (int64_t)b * G;
  const float want = (float)(c + 1);
  int n_cls = 0, first = 0x7fffffff;
  for (int k = lane; k < G; k += 64) {
    const bool is = fb[(int64_t)k * 8 + 7] == want;
    n_cls += is ? 1 : 0;
    if (is && cb[k] > 0) first = min(first, k);
  }
  for (int d = 32; d > 0; d >>= 1) {
    n_cls += __shfl_xor(n_cls, d, 64);
    first = min(first, __shfl_xor(first, d, 64));
  }
  const int dropped = (bg[b * C + c] == 0) ? first : -1;
  int n = 0;
  long long sum = 0;
  for (int k = lane; k < G; k += 64) {
    const bool v = fb[(int64_t)k * 8 + 7] == want && cb[k] > 0 && k != dropped;
    n += v ? 1 : 0;
    sum += v ? cb[k] : 0;
  }
  for (int d = 32; d > 0; d >>= 1) {
    n += __shfl_xor(n, d, 64);
    sum += __shfl_xor(sum, d, 64);
  }
  float* o = stats + (int64_t)blockIdx.x * 5;
  if (n == 0) {
    if (lane == 0) { o[0] = (float)n_cls; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f; o[4] = 0.f; }
    return;
  }
  const double mean_d = (double)sum / (double)n;
  double ss = 0.0;
  int med = -1;
  const int target = (n - 1) / 2;
  for (int k = lane; k < G; k += 64) {
    const bool v = fb[(int64_t)k * 8 + 7] == want && cb[k] > 0 && k != dropped;
    if (!v) continue;
    const int ck = cb[k];
    const double d = (double)ck - mean_d;
    ss += d * d;
    int rank = 0;
    for (int j = 0; j < G; ++j) {
      const bool vj = fb[(int64_t)j * 8 + 7] == want && cb[j] > 0 && j != dropped;
      rank += (vj && (cb[j] < ck || (cb[j] == ck && j < k))) ? 1 : 0;
    }
    if (rank == target) med = ck;
  }
  for (int d = 32; d > 0; d >>= 1) {
    ss += __shfl_xor(ss, d, 64);
    med = max(med, __shfl_xor(med, d, 64));
  }
  if (lane == 0) {
    o[0] = (float)n_cls;
    o[1] = (float)n;
    o[2] = (float)sum / (float)n;
    o[3] = (float)med;
    o[4] = (float)(ss / (double)n);
  }
}

// ---------------------------------------------------------------------------------------------- RoI-aware pool
// one wave per box; pts_idx_of_voxels (N, ox,oy,oz, max_pts) [..,0] = count, then point indices in point order
__global__ __launch_bounds__(256) void roiaware_collect_kernel(int N, int P, int ox, int oy, int oz, int max_pts,
                                                               const float* __restrict__ rois,
                                                               const float* __restrict__ pts,
                                                               int* __restrict__ pts_idx_of_voxels) {
  const int box = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (box >= N) return;
  const BoxLds b = load_box(rois + (int64_t)box * 7);
  int* vox = pts_idx_of_voxels + (int64_t)box * ox * oy * oz * max_pts;
  const float x_res = b.dx / ox, y_res = b.dy / oy, z_res = b.dz / oz;
  const int cap = max_pts - 1;
  for (int k0 = 0; k0 < P; k0 += 64) {
    const int k = k0 + lane;
    int code = -1;
    if (k < P) {
      const float x = pts[(int64_t)k * 3], y = pts[(int64_t)k * 3 + 1], z = pts[(int64_t)k * 3 + 2];
      float lx, ly;
      if (pt_in_box(x, y, z, b, lx, ly)) {
        const float lz = z - b.cz;
        // unsigned conversion + clamp exactly as the reference (negative -> huge unsigned -> clamped to the top cell)
        unsigned xi = (unsigned)(int)((lx + b.dx / 2) / x_res);
        unsigned yi = (unsigned)(int)((ly + b.dy / 2) / y_res);
        unsigned zi = (unsigned)(int)((lz + b.dz / 2) / z_res);
        xi = min(max(xi, 0u), (unsigned)(ox - 1));
        yi = min(max(yi, 0u), (unsigned)(oy - 1));
        zi = min(max(zi, 0u), (unsigned)(oz - 1));
        code = (int)((xi * oy + yi) * oz + zi);
      }
    }
    unsigned long long todo = __ballot(code >= 0);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int c0 = __shfl(code, leader, 64);
      const unsigned long long same = __ballot(code == c0);
      // the group's leader reserves the slots with one returning atomic (ordered at L2: no plain load/store of the
      // counter is ever mixed in), every member places itself by its lane rank -> point order is preserved
      int base = 0;
      if (lane == leader) base = atomicAdd(&vox[(int64_t)c0 * max_pts], __popcll(same));
      base = __shfl(base, leader, 64);
      if (code == c0) {
        const int pos = base + __popcll(same & ((1ULL << lane) - 1ULL));
        if (pos < cap) vox[(int64_t)c0 * max_pts + 1 + pos] = k;
      }
      todo &= ~same;
    }
  }
}

// pooled (N,ox,oy,oz,C), argmax (N,ox,oy,oz,C); one thread per (cell, channel), channel fastest
__global__ __launch_bounds__(256) void roiaware_pool_kernel(int64_t total, int C, int max_pts, int method,
                                                            const float* __restrict__ feat,
                                                            int* __restrict__ pts_idx_of_voxels,
                                                            float* __restrict__ pooled, int* __restrict__ argmax) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int64_t cell = t / C;
  const int c = (int)(t - cell * C);
  int* v = pts_idx_of_voxels + cell * max_pts;
  int n = v[0];
  if (n > max_pts - 1) {               // the collect pass counts every hit; the stored count saturates like the reference
    n = max_pts - 1;
    if (c == 0) v[0] = n;
  }
  if (method == 0) {
    int am = -1;
    float mx = -INFINITY;            // the reference's -1e50 literal stored in a float is -inf
    for (int k = 1; k <= n; ++k) {
      float f = feat[(int64_t)v[k] * C + c];
      if (f > mx) { mx = f; am = v[k]; }
    }
    if (am != -1) pooled[t] = mx;
    argmax[t] = am;
  } else {
    float s = 0.f;
    for (int k = 1; k <= n; ++k) s += feat[(int64_t)v[k] * C + c];
    if (n > 0) pooled[t] = s / n;
  }
}

__global__ __launch_bounds__(256) void roiaware_pool_bwd_kernel(int64_t total, int C, int max_pts, int method,
                                                                const int* __restrict__ pts_idx_of_voxels,
                                                                const int* __restrict__ argmax,
                                                                const float* __restrict__ grad_out,
                                                                float* __restrict__ grad_in) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int64_t cell = t / C;
  const int c = (int)(t - cell * C);
  if (method == 0) {
    const int am = argmax[t];
    if (am == -1) return;
    atomicAdd(&grad_in[(int64_t)am * C + c], grad_out[t] * 1);
  } else {
    const int* v = pts_idx_of_voxels + cell * max_pts;
    const int n = v[0];
    const float g = 1 / fmaxf((float)n, 1.0f);
    for (int k = 1; k <= n; ++k) atomicAdd(&grad_in[(int64_t)v[k] * C + c], grad_out[t] * g);
  }
}

}  // namespace

extern "C" int crb_points_in_boxes(int B, int T, int M, const float* boxes, const float* pts, int32_t* box_idx_of_points,
                                   void* stream) {
  if (B <= 0 || T < 0 || M < 0) return CRB_ERR_ARG;
  if (M == 0) return CRB_OK;
  hipLaunchKernelGGL(points_in_boxes_kernel, dim3(crb_cdiv(M, 256), B), dim3(256), 0, (hipStream_t)stream, T, M, boxes,
                     pts, box_idx_of_points);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int64_t crb_gt_point_stats_workspace_bytes(int B, int G, int C) {
  return (int64_t)sizeof(int) * ((int64_t)B * G + (int64_t)B * C);
}

extern "C" int crb_gt_point_stats(int B, int G, int C, int64_t N, int stride, const float* pts,
                                  const int32_t* frame_offsets, const float* gt_boxes, float* stats, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
  if (B <= 0 || G < 0 || C <= 0 || C > 32 || N < 0 || stride < 4) return CRB_ERR_ARG;
  if (workspace_bytes < crb_gt_point_stats_workspace_bytes(B, G, C) || (workspace == nullptr && G + C > 0))
    return CRB_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int* cnt = (int*)workspace;
  int* bg = cnt + (int64_t)B * G;
  CRB_HIP(hipMemsetAsync(workspace, 0, (size_t)crb_gt_point_stats_workspace_bytes(B, G, C), st));
  if (N > 0 && G > 0)
    hipLaunchKernelGGL(gt_point_count_kernel, dim3(crb_cdiv(N, 256), B), dim3(256), 0, st, G, C, stride, pts,
                       frame_offsets, gt_boxes, cnt, bg);
  hipLaunchKernelGGL(gt_point_stats_kernel, dim3(B * C), dim3(64), 0, st, G, C, gt_boxes, cnt, bg, stats);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_roiaware_pool3d_forward(int N, int P, int C, int max_pts_each_voxel, int out_x, int out_y, int out_z,
                                           const float* rois, const float* pts, const float* pts_feature,
                                           int32_t* argmax, int32_t* pts_idx_of_voxels, float* pooled_features,
                                           int pool_method, void* stream) {
  if (N < 0 || P < 0 || C <= 0 || max_pts_each_voxel < 2 || out_x <= 0 || out_y <= 0 || out_z <= 0) return CRB_ERR_ARG;
  if (out_x > 255 || out_y > 255 || out_z > 255 || pool_method < 0 || pool_method > 1) return CRB_ERR_UNSUPPORTED;
  if (N == 0) return CRB_OK;
  hipStream_t st = (hipStream_t)stream;
  // caller passes pts_idx_of_voxels / pooled_features zero-filled (as the reference's Python side does)
  hipLaunchKernelGGL(roiaware_collect_kernel, dim3(crb_cdiv(N, 4)), dim3(256), 0, st, N, P, out_x, out_y, out_z,
                     max_pts_each_voxel, rois, pts, pts_idx_of_voxels);
  const int64_t total = (int64_t)N * out_x * out_y * out_z * C;
  hipLaunchKernelGGL(roiaware_pool_kernel, dim3(crb_cdiv(total, 256)), dim3(256), 0, st, total, C, max_pts_each_voxel,
                     pool_method, pts_feature, pts_idx_of_voxels, pooled_features, argmax);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_roiaware_pool3d_backward(int N, int C, int max_pts_each_voxel, int out_x, int out_y, int out_z,
                                            const int32_t* pts_idx_of_voxels, const int32_t* argmax,
                                            const float* grad_out, float* grad_in /* pre-zeroed */, int pool_method,
                                            void* stream) {
  if (N < 0 || C <= 0 || pool_method < 0 || pool_method > 1) return CRB_ERR_ARG;
  if (N == 0) return CRB_OK;
  const int64_t total = (int64_t)N * out_x * out_y * out_z * C;
  hipLaunchKernelGGL(roiaware_pool_bwd_kernel, dim3(crb_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, total, C,
                     max_pts_each_voxel, pool_method, pts_idx_of_voxels, argmax, grad_out, grad_in);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}
