// Point head of PV-RCNN in training: point labels behind the two point-in-box queries, and the focal classification loss with its
// gradient, one launch each  (row a20 of SURVEY §8)
//
// replaces, per batch:
//   PointHeadTemplate.assign_stack_targets, the label arithmetic behind points_in_boxes_gpu of the ground truths and of the enlarged
//       ground truths (pcdet/models/dense_heads/point_head_template.py:49-129, PointHeadSimple mode: set_ignore_flag)
//   PointHeadTemplate.get_cls_layer_loss with SigmoidFocalClassificationLoss (:131-155, pcdet/utils/loss_utils.py:9-72) and its autograd
// ~55 elementwise / reduction launches over (32768, <= 4) tensors, ~3 us of device time each in the device-bound PV-RCNN step.
// One workgroup: the positives are counted first (the normaliser), then every term and its derivative; sums in a fixed order.
#include "crb_common.h"
#include "../../include/crb_hip.h"

namespace {

constexpr int TPB = 1024;

__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < TPB / 64; ++w) t += sh[w];
  return t;
}

// focal term of one logit and its derivative w.r.t. the logit, the expressions of rpn_loss.hip (loss_utils.py:47-72)
__device__ __forceinline__ void focal_term(float x, bool t, float alpha, float gamma, float& loss, float& d) {
  const float p = 1.0f / (1.0f + expf(-x));
  const float bce = fmaxf(x, 0.0f) - (t ? x : 0.0f) + log1pf(expf(-fabsf(x)));
  const float pt = t ? 1.0f - p : p;
  const float aw = t ? alpha : 1.0f - alpha;
  float ptg, ptg1;
  if (gamma == 2.0f) {
    ptg = pt * pt;
    ptg1 = 2.0f * pt;
  } else {
    ptg = powf(pt, gamma);
    ptg1 = gamma * powf(pt, gamma - 1.0f);
  }
  loss = aw * ptg * bce;
  const float dpt = t ? -p * (1.0f - p) : p * (1.0f - p);
  const float dbce = x == 0.0f ? (t ? 0.0f : 1.0f) : (t ? p - 1.0f : p);
  d = aw * (ptg1 * dpt * bce + ptg * dbce);
}

__global__ __launch_bounds__(TPB) void point_focal_loss_kernel(const float* __restrict__ preds, const int64_t* __restrict__ labels, int64_t n,
                                                               int C, float alpha, float gamma, float weight, float* __restrict__ out,
                                                               float* __restrict__ d_preds) {
  __shared__ float sh[TPB / 64];
  float np = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += TPB) np += labels[i] > 0 ? 1.f : 0.f;
  const float pos = block_sum(np, sh);                       // (an integer below 2^24: exact in any order)
  const float w = 1.0f / fmaxf(pos, 1.0f);
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += TPB) {
    const int64_t lab = labels[i];
    for (int c = 0; c < C; ++c) {
      float l = 0.f, d = 0.f;
      if (lab >= 0) {                                        // (-1: weight 0)
        focal_term(preds[i * C + c], lab == c + 1, alpha, gamma, l, d);
        l *= w;
        d *= w * weight;
      }
      s += l;
      d_preds[i * C + c] = d;
    }
  }
  const float total = block_sum(s, sh);
  if (threadIdx.x == 0) {
    out[0] = total * weight;
    out[1] = pos;
    out[2] = total * weight;
  }
}

__global__ __launch_bounds__(256) void point_labels_kernel(const int32_t* __restrict__ inner, const int32_t* __restrict__ outer,
                                                           const float* __restrict__ gt, int64_t n, int M, int G, int gt_c, int num_class,
                                                           int64_t* __restrict__ labels) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int in = inner[i];
  const bool fg = in >= 0, ignore = fg != (outer[i] >= 0);
  int64_t lab = ignore ? -1 : 0;
  if (fg) lab = num_class == 1 ? 1 : (int64_t)gt[((i / M) * G + in) * gt_c + gt_c - 1];
  labels[i] = lab;
}

}  // namespace

extern "C" int crb_point_focal_loss(const float* preds, const int64_t* labels, int64_t n, int num_class, float alpha, float gamma,
                                    float loss_weight, float* loss, float* d_preds, void* stream) {
  if (n < 0 || n >= (1LL << 24) || num_class <= 0 || !loss) return CRB_ERR_ARG;
  if (n > 0 && (!preds || !labels || !d_preds)) return CRB_ERR_ARG;
  hipLaunchKernelGGL(point_focal_loss_kernel, dim3(1), dim3(TPB), 0, (hipStream_t)stream, preds, labels, n, num_class, alpha, gamma,
                     loss_weight, loss, d_preds);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_point_labels(const int32_t* inner, const int32_t* outer, const float* gt_boxes, int B, int64_t M, int G,
                                int gt_row_stride, int num_class, int64_t* labels, void* stream) {
  if (B <= 0 || M < 0 || G <= 0 || gt_row_stride < 8 || num_class <= 0 || M >= (1LL << 31)) return CRB_ERR_ARG;
  if (M == 0) return CRB_OK;
  if (!inner || !outer || !gt_boxes || !labels) return CRB_ERR_ARG;
  const int64_t n = (int64_t)B * M;
  hipLaunchKernelGGL(point_labels_kernel, dim3(crb_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, inner, outer, gt_boxes, n, (int)M, G,
                     gt_row_stride, num_class, labels);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}
