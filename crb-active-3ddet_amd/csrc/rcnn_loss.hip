// Second-stage (RCNN) losses of the RoI head as ONE launch, gradients included, and the canonical transformation of the sampled
// ground truths as one launch  (rows a22 / a24 of SURVEY §8)
//
// replaces, for the n = B * ROI_PER_IMAGE sampled RoIs of a batch:
//   RoIHeadTemplate.assign_targets, the part after the sampling   (pcdet/models/roi_heads/roi_head_template.py:118-138)
//   RoIHeadTemplate.get_box_cls_layer_loss, BinaryCrossEntropy    (:261-285)
//   RoIHeadTemplate.get_box_reg_layer_loss, smooth-l1 + corner regularisation (:142-259, the branch without
//       `reg_sample_targets`), with ResidualCoder.encode_torch / decode_torch (pcdet/utils/box_coder_utils.py:13-73),
//       WeightedSmoothL1Loss (pcdet/utils/loss_utils.py:75-131), get_corner_loss_lidar (:209-232),
//       boxes_to_corners_3d (pcdet/utils/box_utils.py:28-55), rotate_points_along_z (pcdet/utils/common_utils.py:37-60)
// which in torch are ~150 elementwise / reduction launches forward and ~120 backward over (2048, <= 24) tensors: at ~3 us of device
// time per launch (tools/bench_pvrcnn.py CRB_BENCH_EXTRA_LAUNCHES) 0.8 ms of a device-bound 68.7 ms PV-RCNN step.
//
// One workgroup of 1024 threads walks the RoIs (2 per thread at n = 2048): counts first (the two denominators), then every term
// and its gradient w.r.t. the head outputs in registers, sums in a fixed order (thread-strided, wave shuffle tree, 16 waves in
// order): bit-reproducible. The gradients are written scaled for d(total loss) = 1; the autograd node multiplies by what arrives.
#include "crb_common.h"
#include "../../include/crb_hip.h"

namespace {

constexpr int TPB = 1024;
constexpr float PI_F = 3.14159265358979323846f;

struct RcnnArgs {
  const float* cls;          // (n) logits
  const float* reg;          // (n, 7)
  const void* labels;        // (n) f32 or i64: < 0 ignored
  const int64_t* reg_valid;  // (n)
  const float* rois;         // (n, 7)
  const float* gt_local;     // (n, gt_stride) canonical (RoI frame) ground truth
  const float* gt_src;       // (n, gt_stride) the same boxes in LiDAR coordinates
  int n, gt_stride, labels_i64, corner;
  float beta, cw[7], w_cls, w_reg, w_corner;
  float* out;                // [0] cls, [1] reg (without corner), [2] corner, [3] total, [4] foreground RoIs, [5] valid RoIs, [6] total
  float* d_cls;              // (n)
  float* d_reg;              // (n, 7)
  float* reg_targets;        // (n, 7) or null
};

// torch.remainder for floats (sign of the divisor)
__device__ __forceinline__ float py_mod(float a, float b) {
  float r = fmodf(a, b);
  if (r != 0.f && ((r < 0.f) != (b < 0.f))) r += b;
  return r;
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
  // wave tree, then the 16 wave sums in wave order by every thread (same value everywhere)
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

// corners of a box (x, y, z, dx, dy, dz, ry): boxes_to_corners_3d's template order
__device__ __forceinline__ void corner_of(int k, float& tx, float& ty, float& tz) {
  tx = (k & 3) == 0 || (k & 3) == 1 ? 0.5f : -0.5f;
  ty = (k & 3) == 0 || (k & 3) == 3 ? 0.5f : -0.5f;
  tz = k < 4 ? -0.5f : 0.5f;
}

__global__ __launch_bounds__(TPB) void rcnn_loss_kernel(RcnnArgs a) {
  __shared__ float sh[TPB / 64];
  // ---- denominators
  float nv = 0.f, nf = 0.f;
  for (int i = threadIdx.x; i < a.n; i += TPB) {
    const float lab = a.labels_i64 ? (float)reinterpret_cast<const int64_t*>(a.labels)[i] : reinterpret_cast<const float*>(a.labels)[i];
    nv += lab >= 0.f ? 1.f : 0.f;
    nf += a.reg_valid[i] > 0 ? 1.f : 0.f;
  }
  const float n_valid = block_sum(nv, sh), n_fg = block_sum(nf, sh);       // (integers below 2^24: exact in any order)
  const float den_v = fmaxf(n_valid, 1.f), den_f = fmaxf(n_fg, 1.f);
  float s_cls = 0.f, s_reg = 0.f, s_cor = 0.f;
  for (int i = threadIdx.x; i < a.n; i += TPB) {
    // ---- classification: binary cross entropy on sigmoid(x) (ATen's kernels: log clamped at -100, backward through
    //      (p - y) / max((1 - p) p, 1e-12) and the sigmoid's (1 - p) p)
    const float lab = a.labels_i64 ? (float)reinterpret_cast<const int64_t*>(a.labels)[i] : reinterpret_cast<const float*>(a.labels)[i];
    float gc = 0.f;
    if (lab >= 0.f) {
      const float x = a.cls[i];
      const float p = 1.0f / (1.0f + expf(-x));
      const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(log1pf(-p), -100.f);
      s_cls += (lab - 1.f) * l1p - lab * lp;
      gc = (p - lab) / fmaxf((1.f - p) * p, 1e-12f) * ((1.f - p) * p) * (a.w_cls / den_v);
    }
    a.d_cls[i] = gc;
    // ---- regression
    const float* e = a.reg + (int64_t)i * 7;
    const float* roi = a.rois + (int64_t)i * 7;
    const float* gl = a.gt_local + (int64_t)i * a.gt_stride;
    const bool fg = a.reg_valid[i] > 0;
    float ge[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float tgt[7];
    {
      // ResidualCoder.encode_torch against the anchor (0, 0, 0, roi dims, 0)
      const float dxa = fmaxf(roi[3], 1e-5f), dya = fmaxf(roi[4], 1e-5f), dza = fmaxf(roi[5], 1e-5f);
      const float dxg = fmaxf(gl[3], 1e-5f), dyg = fmaxf(gl[4], 1e-5f), dzg = fmaxf(gl[5], 1e-5f);
      const float diag = sqrtf(dxa * dxa + dya * dya);
      tgt[0] = gl[0] / diag; tgt[1] = gl[1] / diag; tgt[2] = gl[2] / dza;
      tgt[3] = logf(dxg / dxa); tgt[4] = logf(dyg / dya); tgt[5] = logf(dzg / dza);
      tgt[6] = gl[6];
    }
    if (a.reg_targets) {
#pragma unroll
      for (int j = 0; j < 7; ++j) a.reg_targets[(int64_t)i * 7 + j] = tgt[j];
    }
    if (fg) {
      float row = 0.f;
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        float diff = isnan(tgt[j]) ? 0.f : e[j] - tgt[j];
        diff *= a.cw[j];
        const float m = fabsf(diff);
        float l, g;
        if (a.beta < 1e-5f) {
          l = m;
          g = diff > 0.f ? 1.f : diff < 0.f ? -1.f : 0.f;
        } else if (m < a.beta) {
          l = 0.5f * m * m / a.beta;
          g = diff / a.beta;
        } else {
          l = m - 0.5f * a.beta;
          g = diff > 0.f ? 1.f : -1.f;
        }
        row += l;
        ge[j] = g * a.cw[j] * (a.w_reg / den_f);
      }
      s_reg += row;
      if (a.corner) {
        // decode against (0, 0, 0, roi dims, roi ry), turn the centre by roi ry, move to the RoI centre
        const float* gs = a.gt_src + (int64_t)i * a.gt_stride;
        const float dxa = roi[3], dya = roi[4], dza = roi[5], ra = roi[6];
        const float diag = sqrtf(dxa * dxa + dya * dya);
        const float xl = e[0] * diag, yl = e[1] * diag, zl = e[2] * dza;
        const float dx = expf(e[3]) * dxa, dy = expf(e[4]) * dya, dz = expf(e[5]) * dza;
        const float rg = e[6] + ra;
        const float ca = cosf(ra), sa = sinf(ra);
        const float cx = xl * ca - yl * sa + roi[0], cy = xl * sa + yl * ca + roi[1], cz = zl + roi[2];
        const float cg = cosf(rg), sg = sinf(rg);
        const float c1 = cosf(gs[6]), s1 = sinf(gs[6]);
        const float rf = gs[6] + PI_F;
        const float c2 = cosf(rf), s2 = sinf(rf);
        float tot = 0.f, dcx = 0.f, dcy = 0.f, dcz = 0.f, ddx = 0.f, ddy = 0.f, ddz = 0.f, drg = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float tx, ty, tz;
          corner_of(k, tx, ty, tz);
          const float lx = dx * tx, ly = dy * ty, lz = dz * tz;
          const float px = lx * cg - ly * sg + cx, py = lx * sg + ly * cg + cy, pz = lz + cz;
          const float gx = gs[3] * tx, gy = gs[4] * ty, gz = gs[5] * tz + gs[2];
          const float q1x = gx * c1 - gy * s1 + gs[0], q1y = gx * s1 + gy * c1 + gs[1];
          const float q2x = gx * c2 - gy * s2 + gs[0], q2y = gx * s2 + gy * c2 + gs[1];
          const float u1x = px - q1x, u1y = py - q1y, u2x = px - q2x, u2y = py - q2y, uz = pz - gz;
          const float d1 = sqrtf(u1x * u1x + u1y * u1y + uz * uz), d2 = sqrtf(u2x * u2x + u2y * u2y + uz * uz);
          const float d = fminf(d1, d2);
          tot += d < 1.f ? 0.5f * d * d : d - 0.5f;
          const float sl = (d < 1.f ? d : 1.f) * 0.125f;          // d smooth-l1 / d dist, mean over the 8 corners
          // torch.min gives half of the gradient to each side of a tie; the norm's gradient at 0 is 0
          const float w1 = d1 < d2 ? 1.f : d1 == d2 ? 0.5f : 0.f, w2 = 1.f - w1;
          const float i1 = d1 > 0.f ? sl * w1 / d1 : 0.f, i2 = d2 > 0.f ? sl * w2 / d2 : 0.f;
          const float gpx = i1 * u1x + i2 * u2x, gpy = i1 * u1y + i2 * u2y, gpz = (i1 + i2) * uz;
          dcx += gpx; dcy += gpy; dcz += gpz;
          ddx += gpx * (tx * cg) + gpy * (tx * sg);
          ddy += gpy * (ty * cg) - gpx * (ty * sg);
          ddz += gpz * tz;
          drg += gpx * (-lx * sg - ly * cg) + gpy * (lx * cg - ly * sg);
        }
        s_cor += tot * 0.125f;
        const float sc = a.w_corner / den_f;
        const float dxl = dcx * ca + dcy * sa, dyl = dcy * ca - dcx * sa;
        ge[0] += dxl * diag * sc; ge[1] += dyl * diag * sc; ge[2] += dcz * dza * sc;
        ge[3] += ddx * dx * sc; ge[4] += ddy * dy * sc; ge[5] += ddz * dz * sc;
        ge[6] += drg * sc;
      }
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) a.d_reg[(int64_t)i * 7 + j] = ge[j];
  }
  const float t_cls = block_sum(s_cls, sh), t_reg = block_sum(s_reg, sh), t_cor = block_sum(s_cor, sh);
  if (threadIdx.x == 0) {
    const float l_cls = t_cls / den_v * a.w_cls, l_reg = t_reg / den_f * a.w_reg, l_cor = a.corner ? t_cor / den_f * a.w_corner : 0.f;
    a.out[0] = l_cls;
    a.out[1] = l_reg;
    a.out[2] = l_cor;
    a.out[3] = l_cls + (l_reg + l_cor);
    a.out[4] = n_fg;
    a.out[5] = n_valid;
    a.out[6] = a.out[3];
  }
}

// gt (n, C >= 7) in LiDAR coordinates -> the RoI's frame: centre relative to the RoI and turned by -ry(roi), heading relative to
// the RoI folded into [-pi/2, pi/2] (a box and its 180-degree turn are the same target); columns 3..5 and 7.. pass through
__global__ __launch_bounds__(256) void roi_canonical_kernel(const float* __restrict__ rois, int roi_stride, const float* __restrict__ gt,
                                                            int C, int n, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* r = rois + (int64_t)i * roi_stride;
  const float* g = gt + (int64_t)i * C;
  float* o = out + (int64_t)i * C;
  const float two_pi = 2.f * PI_F;
  const float roi_ry = py_mod(r[6], two_pi);
  const float x = g[0] - r[0], y = g[1] - r[1], z = g[2] - r[2];
  const float ang = -roi_ry;
  const float c = cosf(ang), s = sinf(ang);
  o[0] = x * c - y * s;
  o[1] = x * s + y * c;
  o[2] = z;
  o[3] = g[3]; o[4] = g[4]; o[5] = g[5];
  float h = py_mod(g[6] - roi_ry, two_pi);
  if (h > PI_F * 0.5f && h < PI_F * 1.5f) h = py_mod(h + PI_F, two_pi);
  if (h > PI_F) h = h - PI_F * 2.f;
  h = fminf(fmaxf(h, -PI_F / 2.f), PI_F / 2.f);
  o[6] = h;
  for (int j = 7; j < C; ++j) o[j] = g[j];
}

// ---- ProposalTargetLayer: RoI sampling of one frame per workgroup (thread r = proposal r) ------------------------------------------
// The batched, synchronisation-free form of the mirror (pcdet/models/roi_heads/target_assigner/proposal_target_layer.py here; the
// reference's per-frame loop with np.random / torch.randint: proposal_target_layer.py:93-228 there) as one launch: same-class maximum
// IoU, the three sets (foreground / hard / easy background), quotas, the draws from the given uniforms, every gather, the regression
// mask and the classification labels. ~135 torch launches of a PV-RCNN step.
struct RoiSampleArgs {
  const float* rois; const float* scores; const int64_t* labels; const float* gt; const float* iou;
  const float* u_perm; const float* u_slot;
  int B, R, G, P, roi_c, gt_c;
  CrbRoiSamplerCfg cfg;
  int64_t* sampled; float* o_rois; float* o_gt; float* o_iou; float* o_scores; int64_t* o_labels; int64_t* reg_valid; void* cls_labels;
};

constexpr int RS_MAX_R = 1024;

__device__ __forceinline__ int block_count(bool flag, int* sh) {
  const unsigned long long m = __ballot(flag);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  int t = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
  return t;
}

__global__ __launch_bounds__(RS_MAX_R) void roi_sample_kernel(RoiSampleArgs a) {
  __shared__ float s_mx[RS_MAX_R], s_u[RS_MAX_R];
  __shared__ int s_arg[RS_MAX_R], s_fg[RS_MAX_R], s_hard[RS_MAX_R], s_easy[RS_MAX_R];
  __shared__ unsigned char s_set[RS_MAX_R];                 // bit 0 fg, 1 hard, 2 easy
  __shared__ int s_cnt[RS_MAX_R / 64];
  __shared__ int s_last, s_umin;
  const int b = blockIdx.x, r = threadIdx.x;
  const CrbRoiSamplerCfg& c = a.cfg;
  const bool live = r < a.R;
  // last ground truth of the frame that is not a padding row (every row up to it counts, as in the mirror)
  if (r == 0) s_last = 0;
  __syncthreads();
  if (c.by_class) {
    for (int g = r; g < a.G; g += blockDim.x) {
      const float* row = a.gt + ((int64_t)b * a.G + g) * a.gt_c;
      float sum = 0.f;
      for (int j = 0; j < a.gt_c; ++j) sum += row[j];       // the whole row, class label included (proposal_target_layer.py:93)
      if (sum != 0.f) atomicMax(&s_last, g);
    }
  }
  __syncthreads();
  float mx = 0.f;
  int arg = 0;
  if (live) {
    const float* irow = a.iou + ((int64_t)b * a.R + r) * ((int64_t)a.B * a.G) + (int64_t)b * a.G;
    if (c.by_class) {
      const int64_t lab = a.labels[(int64_t)b * a.R + r];
      float best = -1.f;
      for (int g = 0; g <= s_last && g < a.G; ++g) {
        const bool same = lab == (int64_t)a.gt[((int64_t)b * a.G + g) * a.gt_c + a.gt_c - 1];
        const float v = same ? irow[g] : -1.f;
        if (v > best) { best = v; arg = g; }
      }
      if (best < 0.f) { best = 0.f; arg = 0; }
      mx = best;
    } else {
      mx = irow[0];
      for (int g = 1; g < a.G; ++g)
        if (irow[g] > mx) { mx = irow[g]; arg = g; }
    }
    s_mx[r] = mx;
    s_arg[r] = arg;
    s_u[r] = a.u_perm[(int64_t)b * a.R + r];
  }
  const bool fg = live && mx >= c.fg_thresh;
  const bool easy = live && mx < c.cls_bg_thresh_lo;
  const bool hard = live && mx < c.reg_fg_thresh && mx >= c.cls_bg_thresh_lo;
  if (live) s_set[r] = (fg ? 1 : 0) | (hard ? 2 : 0) | (easy ? 4 : 0);
  const int n_fg = block_count(fg, s_cnt), n_hard = block_count(hard, s_cnt), n_easy = block_count(easy, s_cnt);
  // (after the counts' barriers s_set / s_u / s_mx are visible)
  if (live) {
    // foreground in the order of its uniforms (a random permutation), background sets in proposal order
    int rk_fg = 0, rk_hard = 0, rk_easy = 0, lower = 0;
    const float u = s_u[r];
    for (int q = 0; q < a.R; ++q) {
      const unsigned char sq = s_set[q];
      const float uq = s_u[q];
      const bool before = uq < u || (uq == u && q < r);
      rk_fg += (sq & 1) && before;
      rk_hard += (sq & 2) && q < r;
      rk_easy += (sq & 4) && q < r;
      lower += before;
    }
    if (fg) s_fg[rk_fg] = r;
    if (hard) s_hard[rk_hard] = r;
    if (easy) s_easy[rk_easy] = r;
    if (lower == 0) s_umin = r;
  }
  __syncthreads();
  const int n_bg = n_hard + n_easy, P = a.P;
  int fg_take = n_bg > 0 ? min(n_fg, c.fg_quota) : P;
  if (n_fg == 0) fg_take = 0;
  const int bg_take = P - fg_take;
  int hard_take = n_easy > 0 ? min((int)((float)bg_take * c.hard_bg_ratio), n_hard) : bg_take;
  if (n_hard == 0) hard_take = 0;
  for (int s = r; s < P; s += blockDim.x) {
    const float us = a.u_slot[(int64_t)b * P + s];
    int pick;
    if (s < fg_take) {
      int pos = n_bg > 0 ? s : (int)floorf(us * (float)max(n_fg, 1));
      pos = min(min(pos, a.R - 1), max(n_fg, 1) - 1);
      pick = n_fg > 0 ? s_fg[pos] : s_umin;
    } else if (s < fg_take + hard_take) {
      const int pos = min((int)floorf(us * (float)max(n_hard, 1)), max(n_hard, 1) - 1);
      pick = n_hard > 0 ? s_hard[pos] : 0;
    } else {
      const int pos = min((int)floorf(us * (float)max(n_easy, 1)), max(n_easy, 1) - 1);
      pick = n_easy > 0 ? s_easy[pos] : 0;
    }
    const int64_t o = (int64_t)b * P + s, src = (int64_t)b * a.R + pick;
    a.sampled[o] = pick;
    for (int j = 0; j < a.roi_c; ++j) a.o_rois[o * a.roi_c + j] = a.rois[src * a.roi_c + j];
    const float* grow = a.gt + ((int64_t)b * a.G + s_arg[pick]) * a.gt_c;
    for (int j = 0; j < a.gt_c; ++j) a.o_gt[o * a.gt_c + j] = grow[j];
    const float iou = s_mx[pick];
    a.o_iou[o] = iou;
    a.o_scores[o] = a.scores[src];
    a.o_labels[o] = a.labels[src];
    a.reg_valid[o] = iou > c.reg_fg_thresh ? 1 : 0;
    if (c.score_type == 0) {                                // roi_iou: soft labels
      float v = (iou - c.cls_bg_thresh) / c.soft_den;
      if (iou < c.cls_bg_thresh) v = 0.f;
      if (iou > c.cls_fg_thresh) v = 1.f;
      reinterpret_cast<float*>(a.cls_labels)[o] = v;
    } else {                                                // cls: 1 / 0, -1 between the two thresholds
      int64_t v = iou > c.cls_fg_thresh ? 1 : 0;
      if (iou > c.cls_bg_thresh && iou < c.cls_fg_thresh) v = -1;
      reinterpret_cast<int64_t*>(a.cls_labels)[o] = v;
    }
  }
}

}  // namespace

extern "C" int crb_roi_sample_targets(const float* rois, int roi_row_stride, const float* roi_scores, const int64_t* roi_labels,
                                      const float* gt_boxes, int gt_row_stride, const float* iou, const float* u_perm, const float* u_slot,
                                      int B, int R, int G, const CrbRoiSamplerCfg* cfg, int64_t* sampled, float* out_rois, float* out_gt,
                                      float* out_iou, float* out_scores, int64_t* out_labels, int64_t* reg_valid_mask, void* cls_labels,
                                      void* stream) {
  if (!cfg || B <= 0 || R <= 0 || G <= 0 || roi_row_stride < 7 || gt_row_stride < 8 || cfg->roi_per_image <= 0) return CRB_ERR_ARG;
  if (R > RS_MAX_R) return CRB_ERR_UNSUPPORTED;
  if (!rois || !roi_scores || !roi_labels || !gt_boxes || !iou || !u_perm || !u_slot || !sampled || !out_rois || !out_gt || !out_iou ||
      !out_scores || !out_labels || !reg_valid_mask || !cls_labels)
    return CRB_ERR_ARG;
  RoiSampleArgs a;
  a.rois = rois; a.scores = roi_scores; a.labels = roi_labels; a.gt = gt_boxes; a.iou = iou; a.u_perm = u_perm; a.u_slot = u_slot;
  a.B = B; a.R = R; a.G = G; a.P = cfg->roi_per_image; a.roi_c = roi_row_stride; a.gt_c = gt_row_stride; a.cfg = *cfg;
  a.sampled = sampled; a.o_rois = out_rois; a.o_gt = out_gt; a.o_iou = out_iou; a.o_scores = out_scores; a.o_labels = out_labels;
  a.reg_valid = reg_valid_mask; a.cls_labels = cls_labels;
  const int threads = (int)crb_align_up(R, 64);
  hipLaunchKernelGGL(roi_sample_kernel, dim3(B), dim3(threads), 0, (hipStream_t)stream, a);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_rcnn_loss(const float* rcnn_cls, const float* rcnn_reg, const void* cls_labels, int labels_are_int64,
                             const int64_t* reg_valid_mask, const float* rois, const float* gt_of_rois, const float* gt_of_rois_src,
                             int gt_row_stride, int64_t n, const CrbRcnnLossCfg* cfg, float* loss, float* d_cls, float* d_reg,
                             float* reg_targets, void* stream) {
  if (n < 0 || n >= (1LL << 24) || !cfg || !loss || gt_row_stride < 7) return CRB_ERR_ARG;
  if (n > 0 && (!rcnn_cls || !rcnn_reg || !cls_labels || !reg_valid_mask || !rois || !gt_of_rois || !d_cls || !d_reg)) return CRB_ERR_ARG;
  if (cfg->corner && n > 0 && !gt_of_rois_src) return CRB_ERR_ARG;
  RcnnArgs a;
  a.cls = rcnn_cls; a.reg = rcnn_reg; a.labels = cls_labels; a.reg_valid = reg_valid_mask; a.rois = rois;
  a.gt_local = gt_of_rois; a.gt_src = gt_of_rois_src;
  a.n = (int)n; a.gt_stride = gt_row_stride; a.labels_i64 = labels_are_int64 ? 1 : 0; a.corner = cfg->corner ? 1 : 0;
  a.beta = cfg->beta;
  for (int j = 0; j < 7; ++j) a.cw[j] = cfg->code_weights[j];
  a.w_cls = cfg->cls_weight; a.w_reg = cfg->reg_weight; a.w_corner = cfg->corner_weight;
  a.out = loss; a.d_cls = d_cls; a.d_reg = d_reg; a.reg_targets = reg_targets;
  hipLaunchKernelGGL(rcnn_loss_kernel, dim3(1), dim3(TPB), 0, (hipStream_t)stream, a);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_roi_canonical_targets(const float* rois, int roi_row_stride, const float* gt_of_rois, int gt_row_stride, int64_t n,
                                         float* out, void* stream) {
  if (n < 0 || n >= (1LL << 31) || roi_row_stride < 7 || gt_row_stride < 7) return CRB_ERR_ARG;
  if (n == 0) return CRB_OK;
  if (!rois || !gt_of_rois || !out) return CRB_ERR_ARG;
  hipLaunchKernelGGL(roi_canonical_kernel, dim3(crb_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, rois, roi_row_stride, gt_of_rois,
                     gt_row_stride, (int)n, out);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

// ---- grid points of the RoI-grid pooling (PVRCNNHead.get_global_grid_points_of_roi / get_dense_grid_points, pvrcnn_head.py:116-141):
//      G^3 points per RoI, ((i + 0.5) / G) * size - size / 2 per axis (index order x slowest, z fastest), turned by the RoI's heading
//      about z, moved to its centre. ~17 torch launches as one.
namespace {
__global__ __launch_bounds__(256) void roi_grid_points_kernel(const float* __restrict__ rois, int roi_c, int64_t n, int G,
                                                              float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int g3 = G * G * G;
  if (t >= n * g3) return;
  const int64_t r = t / g3;
  const int q = (int)(t - r * g3);
  const int ix = q / (G * G), iy = (q / G) % G, iz = q % G;
  const float* b = rois + r * roi_c;
  const float gf = (float)G;
  const float lx = ((float)ix + 0.5f) / gf * b[3] - b[3] / 2.f;
  const float ly = ((float)iy + 0.5f) / gf * b[4] - b[4] / 2.f;
  const float lz = ((float)iz + 0.5f) / gf * b[5] - b[5] / 2.f;
  const float c = cosf(b[6]), s = sinf(b[6]);
  out[t * 3 + 0] = lx * c - ly * s + b[0];
  out[t * 3 + 1] = lx * s + ly * c + b[1];
  out[t * 3 + 2] = lz + b[2];
}
}  // namespace

extern "C" int crb_roi_grid_points(const float* rois, int roi_row_stride, int64_t n, int grid_size, float* out, void* stream) {
  if (n < 0 || roi_row_stride < 7 || grid_size <= 0 || grid_size > 64 || n * grid_size * grid_size * grid_size >= (1LL << 40)) return CRB_ERR_ARG;
  if (n == 0) return CRB_OK;
  if (!rois || !out) return CRB_ERR_ARG;
  const int64_t total = n * grid_size * grid_size * grid_size;
  hipLaunchKernelGGL(roi_grid_points_kernel, dim3(crb_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, rois, roi_row_stride, n,
                     grid_size, out);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

// ---- second-stage box decode (RoIHeadTemplate.generate_predicted_boxes, roi_head_template.py:335-359): residuals against the RoI as
//      anchor (centre 0, its dims, its heading), the decoded centre turned by the RoI's heading about z and moved to the RoI's centre.
//      The operations of ResidualCoder.decode_torch / rotate_points_along_z in their order; ~20 torch launches as one.
namespace {
__global__ __launch_bounds__(256) void rcnn_decode_kernel(const float* __restrict__ rois, int roi_c, const float* __restrict__ reg, int64_t n,
                                                          float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* r = rois + i * roi_c;
  const float* e = reg + i * 7;
  const float dxa = r[3], dya = r[4], dza = r[5], ra = r[6];
  const float diag = sqrtf(dxa * dxa + dya * dya);
  const float xl = e[0] * diag + 0.f, yl = e[1] * diag + 0.f, zl = e[2] * dza + 0.f;
  const float c = cosf(ra), s = sinf(ra);
  float* o = out + i * 7;
  o[0] = xl * c - yl * s + r[0];
  o[1] = xl * s + yl * c + r[1];
  o[2] = zl + r[2];
  o[3] = expf(e[3]) * dxa;
  o[4] = expf(e[4]) * dya;
  o[5] = expf(e[5]) * dza;
  o[6] = e[6] + ra;
}

// ---- CRB stage-1 records behind the final NMS (crb_frame_records of the mirror = Detector3DTemplate.post_processing's per-frame
//      selection, detector3d_template.py:190-234, + the label entropy of crb_sampling.py:86-94): gathers of the kept boxes / scores /
//      labels / logits with zero padding, and the Shannon entropy of the predicted-label histogram. One workgroup per frame.
constexpr int REC_MAX_CLASS = 16;
__global__ __launch_bounds__(256) void record_rows_kernel(const int64_t* __restrict__ sel, const uint8_t* __restrict__ valid,
                                                          const float* __restrict__ boxes, int box_c, const float* __restrict__ conf,
                                                          const int64_t* __restrict__ labels, const float* __restrict__ full, int nc_full,
                                                          int N, int P, int num_class, float* __restrict__ o_boxes,
                                                          float* __restrict__ o_scores, int64_t* __restrict__ o_labels,
                                                          float* __restrict__ o_logits, float* __restrict__ o_entropy) {
  __shared__ int cnt[REC_MAX_CLASS];
  __shared__ int nvalid;
  const int b = blockIdx.x;
  if (threadIdx.x < REC_MAX_CLASS) cnt[threadIdx.x] = 0;
  if (threadIdx.x == 0) nvalid = 0;
  __syncthreads();
  for (int j = threadIdx.x; j < P; j += 256) {
    const int64_t o = (int64_t)b * P + j;
    const bool v = valid[o] != 0;
    const float vf = v ? 1.f : 0.f;
    const int64_t src = (int64_t)b * N + sel[o];
    for (int k = 0; k < box_c; ++k) o_boxes[o * box_c + k] = boxes[src * box_c + k] * vf;
    o_scores[o] = conf[src] * vf;
    const int64_t lab = labels[src] * (v ? 1 : 0);
    o_labels[o] = lab;
    if (full)
      for (int k = 0; k < nc_full; ++k) o_logits[o * nc_full + k] = full[src * nc_full + k] * vf;
    if (v) {
      int c = (int)(lab - 1);
      c = c < 0 ? 0 : c;
      if (c < num_class) atomicAdd(&cnt[c], 1);
      atomicAdd(&nvalid, 1);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // absent classes count as 1; proportions over the number of boxes, renormalised (torch.distributions.Categorical)
    const float n = (float)nvalid, den = fmaxf(n, 1.f);
    float props[REC_MAX_CLASS], tot = 0.f;
    for (int c = 0; c < num_class; ++c) {
      props[c] = (cnt[c] > 0 ? (float)cnt[c] : 1.f) / den;
      tot += props[c];
    }
    float ent = 0.f;
    for (int c = 0; c < num_class; ++c) {
      const float p = props[c] / tot;
      ent += p * logf(p);
    }
    o_entropy[b] = nvalid > 0 ? -ent : 0.f;
  }
}

// ---- predicted-box point density: points whose FIRST containing box is k (idx from crb_points_in_boxes) over the box volume
constexpr int DEN_MAX_P = 2048;
__global__ __launch_bounds__(1024) void box_density_kernel(const int32_t* __restrict__ idx, const float* __restrict__ boxes, int box_c,
                                                           const uint8_t* __restrict__ valid, int M, int P, float* __restrict__ density) {
  __shared__ int cnt[DEN_MAX_P];
  const int b = blockIdx.x;
  for (int k = threadIdx.x; k < P; k += 1024) cnt[k] = 0;
  __syncthreads();
  for (int m = threadIdx.x; m < M; m += 1024) {
    const int k = idx[(int64_t)b * M + m];
    if (k >= 0 && k < P) atomicAdd(&cnt[k], 1);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < P; k += 1024) {
    const float* bx = boxes + ((int64_t)b * P + k) * box_c;
    const float vol = bx[3] * bx[4] * bx[5];
    density[(int64_t)b * P + k] = valid[(int64_t)b * P + k] ? (float)cnt[k] / fmaxf(vol, 1e-12f) : 0.f;
  }
}
}  // namespace

extern "C" int crb_rcnn_decode_boxes(const float* rois, int roi_row_stride, const float* box_preds, int64_t n, float* out, void* stream) {
  if (n < 0 || n >= (1LL << 31) || roi_row_stride < 7) return CRB_ERR_ARG;
  if (n == 0) return CRB_OK;
  if (!rois || !box_preds || !out) return CRB_ERR_ARG;
  hipLaunchKernelGGL(rcnn_decode_kernel, dim3(crb_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, rois, roi_row_stride, box_preds, n, out);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_record_rows(const int64_t* sel, const uint8_t* valid, const float* box_preds, int box_row_stride, const float* cls_confs,
                               const int64_t* label_preds, const float* full_cls_scores, int full_classes, int B, int N, int P,
                               int num_class, float* pred_boxes, float* pred_scores, int64_t* pred_labels, float* pred_logits,
                               float* entropy, void* stream) {
  if (B <= 0 || N <= 0 || P <= 0 || box_row_stride < 7 || num_class <= 0 || num_class > REC_MAX_CLASS) return CRB_ERR_ARG;
  if (!sel || !valid || !box_preds || !cls_confs || !label_preds || !pred_boxes || !pred_scores || !pred_labels || !entropy) return CRB_ERR_ARG;
  if (full_cls_scores && (full_classes <= 0 || !pred_logits)) return CRB_ERR_ARG;
  hipLaunchKernelGGL(record_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, sel, valid, box_preds, box_row_stride, cls_confs,
                     label_preds, full_cls_scores, full_classes, N, P, num_class, pred_boxes, pred_scores, pred_labels, pred_logits, entropy);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}

extern "C" int crb_box_point_density(const int32_t* first_box, const float* pred_boxes, int box_row_stride, const uint8_t* valid, int B,
                                     int M, int P, float* density, void* stream) {
  if (B <= 0 || M < 0 || P <= 0 || box_row_stride < 7) return CRB_ERR_ARG;
  if (P > DEN_MAX_P) return CRB_ERR_UNSUPPORTED;
  if ((M > 0 && !first_box) || !pred_boxes || !valid || !density) return CRB_ERR_ARG;
  hipLaunchKernelGGL(box_density_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, first_box, pred_boxes, box_row_stride, valid, M, P,
                     density);
  CRB_CHECK_LAUNCH();
  return CRB_OK;
}
