// ORACLE recipe file (ours): a C-ABI shim over the reference's own boxes_iou_bev_cpu
// (pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp:232-252), compiled together with that source where it lies under
// /root/reference by oracle/build_ref.sh into oracle/_ref/libiou3d_ref.so. No reference source is copied.
#include <torch/extension.h>
#include "iou3d_cpu.h"

extern "C" int ref_boxes_iou_bev_cpu(const float* a, int na, const float* b, int nb, float* out) {
  auto opt = torch::TensorOptions().dtype(torch::kFloat32);
  at::Tensor ta = torch::from_blob(const_cast<float*>(a), {na, 7}, opt);
  at::Tensor tb = torch::from_blob(const_cast<float*>(b), {nb, 7}, opt);
  at::Tensor to = torch::from_blob(out, {na, nb}, opt);
  return boxes_iou_bev_cpu(ta, tb, to);
}
