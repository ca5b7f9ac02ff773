/*
 * ORACLE — test infrastructure only (never linked, imported or called by the product path).
 * CPU restatement of the rotated BEV overlap / IoU and the greedy NMS of the reference:
 *   box_overlap / iou_bev         pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:35-234 (same arithmetic as
 *                                  pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp:60-230, which computes sin/cos in double)
 *   boxes_iou3d                   pcdet/ops/iou3d_nms/iou3d_nms_utils.py:48-81
 *   nms (mask + serial scan)      iou3d_nms_kernel.cu:267-311 + iou3d_nms.cpp:90-136
 *   iou_normal / nms_normal       iou3d_nms_kernel.cu:314-372 + iou3d_nms.cpp:139-185
 * Pinned against the reference's own iou3d_cpu.cpp compiled into oracle/_ref (tests/test_oracle_iou3d.py) and the
 * golden IoU matrix it produced (tests/golden/ref_iou3d.npz).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct { float x, y; } Pt;

static float crs(Pt p1, Pt p2, Pt p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }
static float fmn(float a, float b) { return a > b ? b : a; }
static float fmx(float a, float b) { return a > b ? a : b; }

static int rect_cross(Pt p1, Pt p2, Pt q1, Pt q2) {
  return fmn(p1.x, p2.x) <= fmx(q1.x, q2.x) && fmn(q1.x, q2.x) <= fmx(p1.x, p2.x) &&
         fmn(p1.y, p2.y) <= fmx(q1.y, q2.y) && fmn(q1.y, q2.y) <= fmx(p1.y, p2.y);
}

static int in_box(const float* b, Pt p) {
  const float MARGIN = 1e-2f;
  float c = cosf(-b[6]), s = sinf(-b[6]);
  float rx = (p.x - b[0]) * c + (p.y - b[1]) * (-s);
  float ry = (p.x - b[0]) * s + (p.y - b[1]) * c;
  return (fabsf(rx) < b[3] / 2 + MARGIN && fabsf(ry) < b[4] / 2 + MARGIN);
}

static int isect(Pt p1, Pt p0, Pt q1, Pt q0, Pt* ans) {
  if (!rect_cross(p0, p1, q0, q1)) return 0;
  float s1 = crs(q0, p1, p0), s2 = crs(p1, q1, p0), s3 = crs(p0, q1, q0), s4 = crs(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  float s5 = crs(q1, p1, p0);
  if (fabsf(s5 - s1) > 1e-8f) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}

static void corners(const float* b, Pt* c) {
  float hx = b[3] / 2, hy = b[4] / 2;
  float xs[4] = {b[0] - hx, b[0] + hx, b[0] + hx, b[0] - hx};
  float ys[4] = {b[1] - hy, b[1] - hy, b[1] + hy, b[1] + hy};
  float ca = cosf(b[6]), sa = sinf(b[6]);
  for (int k = 0; k < 4; ++k) {
    c[k].x = (xs[k] - b[0]) * ca + (ys[k] - b[1]) * (-sa) + b[0];
    c[k].y = (xs[k] - b[0]) * sa + (ys[k] - b[1]) * ca + b[1];
  }
  c[4] = c[0];
}

float oracle_box_overlap(const float* a, const float* b) {
  Pt ca[5], cb[5], pts[16], ctr = {0.f, 0.f};
  corners(a, ca);
  corners(b, cb);
  int cnt = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (isect(ca[i + 1], ca[i], cb[j + 1], cb[j], &pts[cnt])) {
        ctr.x = ctr.x + pts[cnt].x; ctr.y = ctr.y + pts[cnt].y;
        ++cnt;
      }
  for (int k = 0; k < 4; ++k) {
    if (in_box(a, cb[k])) { ctr.x = ctr.x + cb[k].x; ctr.y = ctr.y + cb[k].y; pts[cnt++] = cb[k]; }
    if (in_box(b, ca[k])) { ctr.x = ctr.x + ca[k].x; ctr.y = ctr.y + ca[k].y; pts[cnt++] = ca[k]; }
  }
  ctr.x /= cnt; ctr.y /= cnt;
  for (int j = 0; j < cnt - 1; ++j)
    for (int i = 0; i < cnt - j - 1; ++i)
      if (atan2f(pts[i].y - ctr.y, pts[i].x - ctr.x) > atan2f(pts[i + 1].y - ctr.y, pts[i + 1].x - ctr.x)) {
        Pt t = pts[i]; pts[i] = pts[i + 1]; pts[i + 1] = t;
      }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; ++k) {
    Pt u = {pts[k].x - pts[0].x, pts[k].y - pts[0].y}, v = {pts[k + 1].x - pts[0].x, pts[k + 1].y - pts[0].y};
    area += u.x * v.y - u.y * v.x;
  }
  return fabsf(area) / 2.0f;
}

float oracle_iou_bev(const float* a, const float* b) {
  float sa = a[3] * a[4], sb = b[3] * b[4];
  float so = oracle_box_overlap(a, b);
  return so / fmaxf(sa + sb - so, 1e-8f);
}

static float iou_normal(const float* a, const float* b) {
  float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
  float inter = w * h;
  return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, 1e-8f);
}

/* mode 0 overlap, 1 bev iou, 2 iou3d */
void oracle_boxes_pairwise(const float* A, int na, const float* B, int nb, float* out, int mode) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < nb; ++j) {
      const float *a = A + 7 * (size_t)i, *b = B + 7 * (size_t)j;
      float r;
      if (mode == 0) r = oracle_box_overlap(a, b);
      else if (mode == 1) r = oracle_iou_bev(a, b);
      else {
        float ov = oracle_box_overlap(a, b);
        float amax = a[2] + a[5] / 2, amin = a[2] - a[5] / 2, bmax = b[2] + b[5] / 2, bmin = b[2] - b[5] / 2;
        float oh = fmaxf(fminf(amax, bmax) - fmaxf(amin, bmin), 0.f);
        float o3 = ov * oh;
        r = o3 / fmaxf(a[3] * a[4] * a[5] + b[3] * b[4] * b[5] - o3, 1e-6f);
      }
      out[(size_t)i * nb + j] = r;
    }
}

/* greedy NMS over score-sorted boxes; keep gets indices; returns count */
int oracle_nms(const float* boxes, int n, float thresh, int rotated, int* keep) {
  char* removed = (char*)calloc((size_t)(n > 0 ? n : 1), 1);
  int num = 0;
  for (int i = 0; i < n; ++i) {
    if (removed[i]) continue;
    keep[num++] = i;
    for (int j = i + 1; j < n; ++j) {
      if (removed[j]) continue;
      float v = rotated ? oracle_iou_bev(boxes + 7 * (size_t)i, boxes + 7 * (size_t)j)
                        : iou_normal(boxes + 7 * (size_t)i, boxes + 7 * (size_t)j);
      if (v > thresh) removed[j] = 1;
    }
  }
  free(removed);
  return num;
}
