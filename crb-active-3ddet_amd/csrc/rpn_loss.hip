// RPN losses of the anchor head as one forward kernel (+ a per-frame finalize) and one backward kernel  (row a11 of SURVEY §8)
//
// replaces, for one batch of head outputs:
//   AnchorHeadTemplate.get_cls_layer_loss      (pcdet/models/dense_heads/anchor_head_template.py:101-141)
//   AnchorHeadTemplate.add_sin_difference      (:143-152), get_direction_target (:154-166)
//   AnchorHeadTemplate.get_box_reg_layer_loss  (:168-214)
//   SigmoidFocalClassificationLoss.forward     (pcdet/utils/loss_utils.py:9-72)
//   WeightedSmoothL1Loss.forward               (:75-131), WeightedCrossEntropyLoss.forward (:168-188)
// which in torch are ~45 elementwise / reduction launches forward and as many backward over (B, A, .) tensors
// (A = 211,200 anchors per KITTI frame): 2.96 ms per SECOND step at bs = 16 (tools/time_rpn_loss.py).
//
// The three losses are sums over anchors of per-anchor terms divided by the frame's number of positive anchors, so one pass
// computes the un-normalised per-frame sums and the count, and a finalize divides (the reference multiplies every term by
// 1/max(npos,1) before summing: same value up to rounding of the sum). Only POSITIVE anchors (label > 0, ~0.1 %) have a
// regression / direction term: box and direction predictions and regression targets are read for those alone — the forward
// pass streams the labels and the class logits (16 B per anchor), the backward pass writes all three gradients (48 B per
// anchor, zeros for every anchor without a term) through LDS so the stores are whole contiguous runs.
// Sums are taken in a fixed order (per-thread, wave shuffle tree, 4 waves, blocks in index order): bit-reproducible.
#include "crb_common.h"
#include "../../include/crb_hip.h"

namespace {

constexpr int TPB = 256;
constexpr int MAX_NC = 8, MAX_NB = 8;

struct RpnArgs {
  const float* cls;       // (B, A, NC) logits
  const float* box;       // (B, A, 7)
  const float* dir;       // (B, A, NB) or null
  const int32_t* labels;  // (B, A)  -1 ignored, 0 background, c > 0 class
  const float* tgt;       // (B, A, 7) encoded regression targets
  const float* anchors;   // (A, 7)
  int B, A, NC, NB;
  float alpha, gamma, beta, dir_offset;
  float cw[7];
  float w_cls, w_loc, w_dir;
};

// focal term of one logit: loss = aw * pt^gamma * bce, d = d loss / d x      (loss_utils.py:47-72)
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
  // d bce / dx = [x >= 0] - t - sign(x) e^-|x| / (1 + e^-|x|) as autograd differentiates the reference's expression
  // (clamp(x, min=0) - x t + log1p(exp(-|x|))): p - t everywhere except at x == 0 exactly, where sign(0) = 0 leaves 1 - t
  const float dbce = x == 0.0f ? (t ? 0.0f : 1.0f) : (t ? p - 1.0f : p);
  d = aw * (ptg1 * dpt * bce + ptg * dbce);
}

// smooth-L1 of one weighted difference: loss, d loss / d diff      (loss_utils.py:98-107)
__device__ __forceinline__ void smooth_l1(float diff, float beta, float& loss, float& d) {
  const float n = fabsf(diff);
  if (beta < 1e-5f) {
    loss = n;
    d = diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f);
  } else if (n < beta) {
    loss = 0.5f * n * n / beta;
    d = diff / beta;
  } else {
    loss = n - 0.5f * beta;
    d = diff > 0.0f ? 1.0f : -1.0f;
  }
}

// regression term of a positive anchor: sum over the 7 code entries, gradient w.r.t. the 7 predictions
__device__ __forceinline__ float box_term(const float* __restrict__ p, const float* __restrict__ t, const RpnArgs& a,
                                          float* __restrict__ g) {
  float sum = 0.0f;
#pragma unroll
  for (int d = 0; d < 7; ++d) {
    float pv = p[d], tv = t[d], scale = 1.0f;
    if (d == 6) {                                  // sin(a - b) = sin a cos b - cos a sin b, as the reference encodes it
      const float sp = sinf(p[6]), cp = cosf(p[6]), st = sinf(t[6]), ct = cosf(t[6]);
      pv = sp * ct;
      tv = cp * st;
      scale = cp * ct + sp * st;                   // d (pv - tv) / d p6
    }
    float diff = (tv != tv) ? 0.0f : (pv - tv);    // NaN target -> target := input (loss_utils.py:119)
    if (tv != tv) scale = 0.0f;
    diff *= a.cw[d];
    float l, dl;
    smooth_l1(diff, a.beta, l, dl);
    sum += l;
    if (g) g[d] = dl * a.cw[d] * scale;
  }
  return sum;
}

// direction bin of a positive anchor      (anchor_head_template.py:154-166, common_utils.limit_period)
__device__ __forceinline__ int dir_bin(float t6, float anchor_rot, const RpnArgs& a) {
  const float two_pi = 6.283185307179586f;
  const float rot_gt = t6 + anchor_rot;
  const float val = rot_gt - a.dir_offset;
  const float off = val - floorf(val / two_pi + 0.0f) * two_pi;
  int bin = (int)floorf(off / (two_pi / (float)a.NB));
  return min(max(bin, 0), a.NB - 1);
}

// cross entropy of NB logits with target bin: loss, gradient = softmax - onehot
__device__ __forceinline__ float dir_term(const float* __restrict__ x, int nb, int bin, float* __restrict__ g) {
  float m = x[0];
  for (int k = 1; k < nb; ++k) m = fmaxf(m, x[k]);
  float s = 0.0f;
  for (int k = 0; k < nb; ++k) s += expf(x[k] - m);
  const float lse = logf(s);
  if (g)
    for (int k = 0; k < nb; ++k) g[k] = expf(x[k] - m - lse) - (k == bin ? 1.0f : 0.0f);
  return -(x[bin] - m - lse);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

constexpr int FWD_PER_THREAD = 4;      // anchors per thread of the forward pass (strided by the block: coalesced label loads)

// partial (B, nblk, 4) = {cls, loc, dir, npos} un-normalised sums of the block's anchors
__global__ __launch_bounds__(TPB) void rpn_loss_partial_kernel(RpnArgs a, float* __restrict__ partial) {
  const int b = blockIdx.y, nblk = gridDim.x;
  const int64_t fb = (int64_t)b * a.A;
  float s_cls = 0.0f, s_loc = 0.0f, s_dir = 0.0f, s_pos = 0.0f;
#pragma unroll
  for (int it = 0; it < FWD_PER_THREAD; ++it) {
    const int i = (blockIdx.x * FWD_PER_THREAD + it) * TPB + threadIdx.x;
    if (i >= a.A) break;
    const int lab = a.labels[fb + i];
    if (lab >= 0) {
      const int tc = lab > 0 ? (a.NC == 1 ? 1 : lab) : 0;
      const float* x = a.cls + (fb + i) * a.NC;
      for (int c = 0; c < a.NC; ++c) {
        float l, d;
        focal_term(x[c], tc == c + 1, a.alpha, a.gamma, l, d);
        s_cls += l;
      }
    }
    if (lab > 0) {
      s_pos += 1.0f;
      const float* t = a.tgt + (fb + i) * 7;
      s_loc += box_term(a.box + (fb + i) * 7, t, a, nullptr);
      if (a.dir) s_dir += dir_term(a.dir + (fb + i) * a.NB, a.NB, dir_bin(t[6], a.anchors[(int64_t)i * 7 + 6], a), nullptr);
    }
  }
  __shared__ float red[TPB / 64][4];
  s_cls = wave_sum(s_cls);
  s_loc = wave_sum(s_loc);
  s_dir = wave_sum(s_dir);
  s_pos = wave_sum(s_pos);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[w][0] = s_cls;
    red[w][1] = s_loc;
    red[w][2] = s_dir;
    red[w][3] = s_pos;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float v = 0.0f;
#pragma unroll
    for (int k = 0; k < TPB / 64; ++k) v += red[k][threadIdx.x];
    partial[((int64_t)b * nblk + blockIdx.x) * 4 + threadIdx.x] = v;
  }
}

// one block per frame: partials summed in block order -> loss (B,3) weighted and normalised, npos (B)
__global__ __launch_bounds__(TPB) void rpn_loss_finalize_kernel(const float* __restrict__ partial, int nblk, float w_cls,
                                                                float w_loc, float w_dir, float* __restrict__ loss,
                                                                float* __restrict__ npos) {
  const int b = blockIdx.x;
  // thread t sums the partials t, t+256, ... (fixed order), then the fixed tree over the 256 threads
  float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int k = threadIdx.x; k < nblk; k += TPB) {
    const float4 p = *reinterpret_cast<const float4*>(partial + ((int64_t)b * nblk + k) * 4);
    v[0] += p.x;
    v[1] += p.y;
    v[2] += p.z;
    v[3] += p.w;
  }
  __shared__ float red[TPB / 64][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = wave_sum(v[j]);
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[threadIdx.x >> 6][j] = v[j];
  __syncthreads();
  if (threadIdx.x == 0) {
    float s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[j] = 0.0f;
      for (int k = 0; k < TPB / 64; ++k) s[j] += red[k][j];
    }
    const float norm = fmaxf(s[3], 1.0f);
    loss[b * 3 + 0] = s[0] / norm * w_cls;
    loss[b * 3 + 1] = s[1] / norm * w_loc;
    loss[b * 3 + 2] = s[2] / norm * w_dir;
    npos[b] = s[3];
  }
}

// gradients of sum_b sum_k gout[b,k] * loss[b,k] w.r.t. the three prediction tensors; a block = 256 consecutive anchors of a
// frame, gradients staged in LDS and stored as contiguous runs
__global__ __launch_bounds__(TPB) void rpn_loss_backward_kernel(RpnArgs a, const float* __restrict__ npos,
                                                                const float* __restrict__ gout, float* __restrict__ dcls,
                                                                float* __restrict__ dbox, float* __restrict__ ddir) {
  __shared__ float s_cls[TPB * MAX_NC];
  __shared__ float s_box[TPB * 7];
  __shared__ float s_dir[TPB * MAX_NB];
  const int b = blockIdx.y;
  const int64_t fb = (int64_t)b * a.A;
  const int i0 = blockIdx.x * TPB, i = i0 + threadIdx.x;
  const int cnt = min(TPB, a.A - i0);
  const float inv = 1.0f / fmaxf(npos[b], 1.0f);
  const float g_cls = gout[b * 3 + 0] * a.w_cls * inv, g_loc = gout[b * 3 + 1] * a.w_loc * inv,
              g_dir = gout[b * 3 + 2] * a.w_dir * inv;
  if (i < a.A) {
    const int lab = a.labels[fb + i];
    const int tc = lab > 0 ? (a.NC == 1 ? 1 : lab) : 0;
    const float* x = a.cls + (fb + i) * a.NC;
    for (int c = 0; c < a.NC; ++c) {
      float l, d = 0.0f;
      if (lab >= 0) focal_term(x[c], tc == c + 1, a.alpha, a.gamma, l, d);
      s_cls[threadIdx.x * a.NC + c] = lab >= 0 ? d * g_cls : 0.0f;
    }
    float gb[7] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    float gd[MAX_NB];
#pragma unroll
    for (int k = 0; k < MAX_NB; ++k) gd[k] = 0.0f;
    if (lab > 0) {
      const float* t = a.tgt + (fb + i) * 7;
      box_term(a.box + (fb + i) * 7, t, a, gb);
      if (a.dir) dir_term(a.dir + (fb + i) * a.NB, a.NB, dir_bin(t[6], a.anchors[(int64_t)i * 7 + 6], a), gd);
    }
#pragma unroll
    for (int d = 0; d < 7; ++d) s_box[threadIdx.x * 7 + d] = gb[d] * g_loc;
    if (a.dir)
#pragma unroll
      for (int k = 0; k < MAX_NB; ++k)
        if (k < a.NB) s_dir[threadIdx.x * a.NB + k] = gd[k] * g_dir;
  }
  __syncthreads();
  float* oc = dcls + (fb + i0) * a.NC;
  for (int k = threadIdx.x; k < cnt * a.NC; k += TPB) oc[k] = s_cls[k];
  float* ob = dbox + (fb + i0) * 7;
  for (int k = threadIdx.x; k < cnt * 7; k += TPB) ob[k] = s_box[k];
  if (a.dir) {
    float* od = ddir + (fb + i0) * a.NB;
    for (int k = threadIdx.x; k < cnt * a.NB; k += TPB) od[k] = s_dir[k];
  }
}

bool fill_args(RpnArgs& a, const float* cls, const float* box, const float* dir, const int32_t* labels, const float* tgt,
               const float* anchors, int B, int A, const CrbRpnLossCfg* cfg) {
  if (!cls || !box || !labels || !tgt || !cfg || B < 0 || A < 0) return false;
  if (cfg->num_class < 1 || cfg->num_class > MAX_NC) return false;
  if (dir && (cfg->num_dir_bins < 1 || cfg->num_dir_bins > MAX_NB || !anchors)) return false;
  a.cls = cls;
  a.box = box;
  a.dir = dir;
  a.labels = labels;
  a.tgt = tgt;
  a.anchors = anchors;
  a.B = B;
  a.A = A;
  a.NC = cfg->num_class;
  a.NB = dir ? cfg->num_dir_bins : 1;
  a.alpha = cfg->alpha;
  a.gamma = cfg->gamma;
  a.beta = cfg->beta;
  a.dir_offset = cfg->dir_offset;
  for (int d = 0; d < 7; ++d) a.cw[d] = cfg->code_weights[d];
  a.w_cls = cfg->cls_weight;
  a.w_loc = cfg->loc_weight;
  a.w_dir = cfg->dir_weight;
  return true;
}

int fwd_blocks(int A) { return (A + TPB * FWD_PER_THREAD - 1) / (TPB * FWD_PER_THREAD); }

}  // namespace

extern "C" {

int64_t crb_rpn_loss_workspace_bytes(int B, int A) {
  if (B <= 0 || A <= 0) return 16;
  return (int64_t)B * fwd_blocks(A) * 4 * sizeof(float);
}

int crb_rpn_loss_forward(const float* cls_preds, const float* box_preds, const float* dir_preds, const int32_t* labels,
                         const float* reg_targets, const float* anchors, int B, int A, const CrbRpnLossCfg* cfg,
                         float* loss, float* npos, void* workspace, int64_t workspace_bytes, void* stream) {
  RpnArgs a;
  if (!fill_args(a, cls_preds, box_preds, dir_preds, labels, reg_targets, anchors, B, A, cfg) || !loss || !npos)
    return CRB_ERR_ARG;
  if (B == 0) return CRB_OK;
  if (A == 0) return CRB_ERR_ARG;
  if (!workspace || workspace_bytes < crb_rpn_loss_workspace_bytes(B, A)) return CRB_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int nblk = fwd_blocks(A);
  float* partial = (float*)workspace;
  hipLaunchKernelGGL(rpn_loss_partial_kernel, dim3(nblk, B), dim3(TPB), 0, s, a, partial);
  hipLaunchKernelGGL(rpn_loss_finalize_kernel, dim3(B), dim3(TPB), 0, s, partial, nblk, a.w_cls, a.w_loc,
                     dir_preds ? a.w_dir : 0.0f, loss, npos);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

int crb_rpn_loss_backward(const float* cls_preds, const float* box_preds, const float* dir_preds, const int32_t* labels,
                          const float* reg_targets, const float* anchors, int B, int A, const CrbRpnLossCfg* cfg,
                          const float* npos, const float* grad_loss, float* d_cls, float* d_box, float* d_dir, void* stream) {
  RpnArgs a;
  if (!fill_args(a, cls_preds, box_preds, dir_preds, labels, reg_targets, anchors, B, A, cfg) || !npos || !grad_loss ||
      !d_cls || !d_box || (dir_preds && !d_dir))
    return CRB_ERR_ARG;
  if (B == 0 || A == 0) return CRB_OK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(rpn_loss_backward_kernel, dim3((A + TPB - 1) / TPB, B), dim3(TPB), 0, s, a, npos, grad_loss, d_cls, d_box,
                     d_dir);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

}  // extern "C"
