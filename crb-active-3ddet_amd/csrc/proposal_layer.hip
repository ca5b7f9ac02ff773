// Proposal layer of the two-stage detectors around its NMS: the decode of the top-k anchors it keeps and the gathers behind the NMS,
// one launch each  (row a21 of SURVEY §8)
//
// replaces, per batch:
//   AnchorHeadTemplate.generate_predicted_boxes restricted to the anchors the proposal layer asks for
//       (pcdet/models/dense_heads/anchor_head_template.py:238-285: ResidualCoder.decode_torch, pcdet/utils/box_coder_utils.py:45-73,
//        direction-bin correction with common_utils.limit_period, pcdet/utils/common_utils.py:24-27)
//   the selection arithmetic of RoIHeadTemplate.proposal_layer after class-agnostic NMS (pcdet/models/roi_heads/roi_head_template.py:
//       73-108: rois / roi_scores / roi_labels of the kept boxes, zero padding)
// ~45 elementwise / gather launches over (B, 9000, .) tensors in training, (B, 1024, .) in the scoring pass.
// Every operation is the f32 operation of the torch expression in the same order (-ffp-contract=off): results equal it bit for bit.
#include "crb_common.h"
#include "../../include/crb_hip.h"

namespace {

__global__ __launch_bounds__(256) void decode_selected_kernel(const float* __restrict__ box, const float* __restrict__ dir,
                                                              const float* __restrict__ anchors, const int64_t* __restrict__ idx, int B,
                                                              int64_t A, int k, int nb, float dir_offset, float dir_limit_offset,
                                                              float period, float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)B * k) return;
  const int b = (int)(t / k);
  const int64_t ai = idx[t];
  const float* a = anchors + ai * 7;
  const float* e = box + ((int64_t)b * A + ai) * 7;
  const float diag = sqrtf(a[3] * a[3] + a[4] * a[4]);
  float* o = out + t * 7;
  o[0] = e[0] * diag + a[0];
  o[1] = e[1] * diag + a[1];
  o[2] = e[2] * a[5] + a[2];
  o[3] = expf(e[3]) * a[3];
  o[4] = expf(e[4]) * a[4];
  o[5] = expf(e[5]) * a[5];
  float r = e[6] + a[6];
  if (dir) {
    const float* d = dir + ((int64_t)b * A + ai) * nb;
    int lab = 0;
    float best = d[0];
    for (int j = 1; j < nb; ++j)
      if (d[j] > best) { best = d[j]; lab = j; }
    const float v = r - dir_offset;
    const float dir_rot = v - floorf(v / period + dir_limit_offset) * period;
    r = dir_rot + dir_offset + period * (float)lab;
  }
  o[6] = r;
}

__global__ __launch_bounds__(256) void proposal_finish_kernel(const int32_t* __restrict__ keep, const int64_t* __restrict__ top_idx,
                                                              const float* __restrict__ top_boxes, const float* __restrict__ scores,
                                                              const int64_t* __restrict__ labels, const float* __restrict__ cls, int B,
                                                              int64_t A, int k, int post, int box_c, int nc, float* __restrict__ rois,
                                                              float* __restrict__ roi_scores, int64_t* __restrict__ roi_labels,
                                                              float* __restrict__ full) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)B * post) return;
  const int b = (int)(t / post);
  const int kp = keep[t];
  const bool valid = kp >= 0;
  const int kc = valid ? kp : 0;
  const float vf = valid ? 1.f : 0.f;
  const int64_t sel = top_idx[(int64_t)b * k + kc];
  for (int j = 0; j < box_c; ++j) rois[t * box_c + j] = top_boxes[((int64_t)b * k + kc) * box_c + j] * vf;
  roi_scores[t] = scores[(int64_t)b * A + sel] * vf;
  roi_labels[t] = labels[(int64_t)b * A + sel] * (valid ? 1 : 0) + 1;
  for (int j = 0; j < nc; ++j) full[t * nc + j] = cls[((int64_t)b * A + sel) * nc + j] * vf;
}

}  // namespace

extern "C" int crb_decode_selected_anchors(const float* box_preds, const float* dir_preds, const float* anchors, const int64_t* anchor_idx,
                                           int B, int64_t A, int k, int num_dir_bins, float dir_offset, float dir_limit_offset, float* out,
                                           void* stream) {
  if (B <= 0 || A <= 0 || k < 0 || (dir_preds && num_dir_bins <= 0)) return CRB_ERR_ARG;
  if (k == 0) return CRB_OK;
  if (!box_preds || !anchors || !anchor_idx || !out) return CRB_ERR_ARG;
  const float period = dir_preds ? (float)(2.0 * 3.14159265358979323846 / (double)num_dir_bins) : 0.f;
  hipLaunchKernelGGL(decode_selected_kernel, dim3(crb_cdiv((int64_t)B * k, 256)), dim3(256), 0, (hipStream_t)stream, box_preds, dir_preds,
                     anchors, anchor_idx, B, A, k, num_dir_bins, dir_offset, dir_limit_offset, period, out);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_proposal_finish(const int32_t* keep, const int64_t* top_idx, const float* top_boxes, const float* scores,
                                   const int64_t* labels, const float* cls_preds, int B, int64_t A, int k, int post, int box_row_stride,
                                   int num_class, float* rois, float* roi_scores, int64_t* roi_labels, float* full_cls_scores, void* stream) {
  if (B <= 0 || A <= 0 || k <= 0 || post <= 0 || box_row_stride < 7 || num_class <= 0) return CRB_ERR_ARG;
  if (!keep || !top_idx || !top_boxes || !scores || !labels || !cls_preds || !rois || !roi_scores || !roi_labels || !full_cls_scores)
    return CRB_ERR_ARG;
  hipLaunchKernelGGL(proposal_finish_kernel, dim3(crb_cdiv((int64_t)B * post, 256)), dim3(256), 0, (hipStream_t)stream, keep, top_idx,
                     top_boxes, scores, labels, cls_preds, B, A, k, post, box_row_stride, num_class, rois, roi_scores, roi_labels,
                     full_cls_scores);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}
