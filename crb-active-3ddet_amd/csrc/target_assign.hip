// Anchor target assignment for gfx950 (row a10 of SURVEY §8): nearest-BEV IoU, arg-max both ways, thresholds and box
// encoding fused into two launches for the whole batch and all anchor classes.
//
// Replaces AxisAlignedTargetAssigner.assign_targets[_single] (pcdet/models/dense_heads/target_assigner/
// axis_aligned_target_assigner.py:36-210) + box_utils.boxes3d_nearest_bev_iou (pcdet/utils/box_utils.py:272-298) +
// ResidualCoder.encode_torch (pcdet/utils/box_coder_utils.py:13-43): per frame x class Python loops over a
// (70400, g) IoU matrix with nonzero()/.cpu() round trips there; the host mirror's batched torch version still
// materialises three (B, 70400, G) tensors and ~40 elementwise/reduce launches per step.
//
// The arithmetic follows the PyTorch ops of the reference one for one in f32 (limit_period as v - floor(v*(1/pi) + 0.5)*pi
// — ATen multiplies by the reciprocal when dividing by a scalar —, IoU = inter / max(area_a + area_b - inter, 1e-6),
// thresholds compared in f32), so labels are identical to the torch path; regression targets use the same formulas.
//
// pass 1: per (frame, anchor): IoU against the frame's ground truths of the anchor's class -> best gt (first max),
//         and atomicMax of the IoU bit pattern into g2a_max[frame][gt] (IoU >= 0, so int order == float order)
// pass 2: per (frame, anchor): forced = any gt with IoU == g2a_max[gt] > 0; label / target / weight.
#include "crb_common.h"
#include "../../include/crb_hip.h"

namespace {

constexpr int MAXG = 1024;    // ground truths per frame staged in LDS

struct Aligned { float x1, y1, x2, y2; };

__device__ __forceinline__ Aligned aligned_bev(float x, float y, float dx, float dy, float heading) {
  const float pi = 3.14159274101257324f;                 // float(np.pi)
  const float inv_pi = 1.0f / pi;
  float r = heading - floorf(heading * inv_pi + 0.5f) * pi;
  r = fabsf(r);
  const float quarter = 0.785398185253143310546875f;     // float(np.pi / 4)
  const float w = (r < quarter) ? dx : dy;
  const float h = (r < quarter) ? dy : dx;
  Aligned a;
  a.x1 = x - w / 2; a.y1 = y - h / 2; a.x2 = x + w / 2; a.y2 = y + h / 2;
  return a;
}

__device__ __forceinline__ float iou_aligned(const Aligned& a, const Aligned& b) {
  const float xl = fmaxf(fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1), 0.f);
  const float yl = fmaxf(fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1), 0.f);
  const float area_a = (a.x2 - a.x1) * (a.y2 - a.y1);
  const float area_b = (b.x2 - b.x1) * (b.y2 - b.y1);
  const float inter = xl * yl;
  return inter / fmaxf(area_a + area_b - inter, 1e-6f);
}

// anchors (A,7) in the flattened (z,y,x,class,size,rot) order, anchor_cls (A) i32 1-based class of every anchor
// gt (B,G,8) zero padded; gvalid (B,G) u8: rows up to the last non-zero row
template <int PASS>
__global__ __launch_bounds__(256) void assign_kernel(const float* __restrict__ anchors, const int* __restrict__ anchor_cls,
                                                     int A, const float* __restrict__ gt, const unsigned char* __restrict__ gvalid,
                                                     int G, const float* __restrict__ matched, const float* __restrict__ unmatched,
                                                     int* __restrict__ g2a_max /* (B,G) IoU bits */,
                                                     int* __restrict__ best_gt /* (B,A) */, float* __restrict__ best_iou,
                                                     int* __restrict__ labels, float* __restrict__ targets,
                                                     float* __restrict__ reg_weights) {
  __shared__ Aligned sg[MAXG];
  __shared__ int scls[MAXG];
  __shared__ int smax[MAXG];       // pass 1: this block's running max per gt; pass 2: the global max
  __shared__ int has_cls[8];
  const int b = blockIdx.y;
  const float* fg = gt + (int64_t)b * G * 8;
  if (threadIdx.x < 8) has_cls[threadIdx.x] = 0;
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += 256) {
    const float* q = fg + g * 8;
    sg[g] = aligned_bev(q[0], q[1], q[3], q[4], q[6]);
    const int c = gvalid[(int64_t)b * G + g] ? (int)q[7] : 0;
    scls[g] = c;
    if (c > 0 && c < 8) has_cls[c] = 1;
    smax[g] = (PASS == 2) ? g2a_max[(int64_t)b * G + g] : 0;
  }
  __syncthreads();
  const int a = blockIdx.x * 256 + threadIdx.x;
  const bool live = a < A;
  if (PASS == 2 && !live) return;
  const float* an = anchors + (int64_t)(live ? a : 0) * 7;
  const int cls = anchor_cls[live ? a : 0];
  const Aligned ab = aligned_bev(an[0], an[1], an[3], an[4], an[6]);
  const int64_t o = (int64_t)b * A + a;
  if (PASS == 1) {
    // per-gt best overlap: LDS max per block first (only non-zero overlaps matter: the table starts at 0), then one
    // global atomic per (block, gt) — one global atomic per (anchor, gt) serialised 70k threads on ~12 addresses
    if (live) {
      float best = -1.f;
      int arg = 0;
      for (int g = 0; g < G; ++g) {
        if (scls[g] != cls) continue;
        const float v = iou_aligned(ab, sg[g]);
        if (v > best) { best = v; arg = g; }
        if (v > 0.f) atomicMax(&smax[g], __float_as_int(v));
      }
      best_gt[o] = arg;
      best_iou[o] = best;
    }
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += 256)
      if (smax[g] > 0) atomicMax(&g2a_max[(int64_t)b * G + g], smax[g]);
  } else {
    const float best = best_iou[o];
    const int arg = best_gt[o];
    bool forced = false;
    for (int g = 0; g < G; ++g) {
      if (scls[g] != cls) continue;
      const int mx = smax[g];
      if (mx <= 0) continue;                                   // best overlap 0: the reference sets it to -1 (never equal)
      forced |= (__float_as_int(iou_aligned(ab, sg[g])) == mx);
    }
    const bool any_gt = has_cls[cls & 7] != 0;
    const bool fgf = any_gt && (forced || best >= matched[cls]);
    const bool bgf = (best < unmatched[cls]) && !forced;
    int lab = fgf ? cls : (bgf ? 0 : -1);
    if (!any_gt) lab = 0;
    labels[o] = lab;
    reg_weights[o] = fgf ? 1.0f : 0.0f;
    float* t = targets + o * 7;
    if (fgf) {
      const float* q = fg + arg * 8;
      const float dxa = fmaxf(an[3], 1e-5f), dya = fmaxf(an[4], 1e-5f), dza = fmaxf(an[5], 1e-5f);
      const float dxg = fmaxf(q[3], 1e-5f), dyg = fmaxf(q[4], 1e-5f), dzg = fmaxf(q[5], 1e-5f);
      const float diag = sqrtf(dxa * dxa + dya * dya);
      t[0] = (q[0] - an[0]) / diag;
      t[1] = (q[1] - an[1]) / diag;
      t[2] = (q[2] - an[2]) / dza;
      t[3] = logf(dxg / dxa);
      t[4] = logf(dyg / dya);
      t[5] = logf(dzg / dza);
      t[6] = q[6] - an[6];
    } else {
#pragma unroll
      for (int k = 0; k < 7; ++k) t[k] = 0.f;
    }
  }
}

}  // namespace

extern "C" int64_t crb_assign_targets_workspace_bytes(int B, int A, int G) {
  return crb_align_up((int64_t)B * G * 4, 256) + crb_align_up((int64_t)B * A * 4, 256) * 2 + 256;
}

extern "C" int crb_assign_targets(const float* anchors, const int32_t* anchor_cls, int A, const float* gt_boxes,
                                  const uint8_t* gt_valid, int B, int G, const float* matched_thr,
                                  const float* unmatched_thr, int32_t* labels, float* reg_targets, float* reg_weights,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
  if (A <= 0 || B <= 0 || G <= 0) return CRB_ERR_ARG;
  if (G > MAXG) return CRB_ERR_UNSUPPORTED;
  CrbArena a(workspace, (size_t)workspace_bytes);
  int* g2a = a.take<int>((int64_t)B * G);
  int* best_gt = a.take<int>((int64_t)B * A);
  float* best_iou = a.take<float>((int64_t)B * A);
  if (!a.ok) return CRB_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  CRB_HIP(hipMemsetAsync(g2a, 0, sizeof(int) * (size_t)B * G, st));
  dim3 grid(crb_cdiv(A, 256), B);
  hipLaunchKernelGGL(assign_kernel<1>, grid, dim3(256), 0, st, anchors, anchor_cls, A, gt_boxes, gt_valid, G, matched_thr,
                     unmatched_thr, g2a, best_gt, best_iou, labels, reg_targets, reg_weights);
  hipLaunchKernelGGL(assign_kernel<2>, grid, dim3(256), 0, st, anchors, anchor_cls, A, gt_boxes, gt_valid, G, matched_thr,
                     unmatched_thr, g2a, best_gt, best_iou, labels, reg_targets, reg_weights);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}
