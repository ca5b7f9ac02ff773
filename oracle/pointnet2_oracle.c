/*
 * ORACLE — test infrastructure only (never linked, imported or called by the product path).
 * CPU restatement, loop for loop, of the reference's PointNet++ stack kernels and point/box ops:
 *   ball_query_kernel_stack            pcdet/ops/pointnet2/pointnet2_stack/src/ball_query_gpu.cu:16-66
 *   group_points[_grad]_kernel_stack   .../group_points_gpu.cu:15-102
 *   farthest_point_sampling_kernel     .../sampling_gpu.cu:25-140  (incl. its strided-thread + LDS-tree tie rule)
 *   three_nn / three_interpolate[_grad] .../interpolate_gpu.cu:16-172
 *   points_in_boxes_kernel             pcdet/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:23-36,313-336
 *   roiaware pool (mask, collect, max/avg, backward)  .../roiaware_pool3d_kernel.cu:39-190,236-290
 * PARITY UNPINNED by the reference for the pointnet2 ops and the RoI-aware pool (no tests; the .cu files cannot run
 * without a GPU). The points-in-box rotation + extent test IS pinned: bit-exact against the reference's own
 * points_in_boxes_cpu compiled from /root/reference (oracle/_ref/roiaware_pool3d_ref*.so) + golden ref_points_in_boxes.npz.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int batch_of(const int* cnt, int B, int i) {
  int b = 0, acc = cnt[0];
  for (int k = 1; k < B; ++k) { if (i < acc) break; acc += cnt[k]; b = k; }
  return b;
}
static int start_of(const int* cnt, int b) { int s = 0; for (int k = 0; k < b; ++k) s += cnt[k]; return s; }

void oracle_ball_query(int B, int M, float radius, int nsample, const float* new_xyz, const int* new_cnt,
                       const float* xyz, const int* xyz_cnt, int* idx /* (M,nsample) zero-initialised by us */) {
  memset(idx, 0, sizeof(int) * (size_t)M * nsample);
  for (int q = 0; q < M; ++q) {
    int b = batch_of(new_cnt, B, q);
    const float* p = xyz + 3 * (size_t)start_of(xyz_cnt, b);
    int n = xyz_cnt[b];
    float r2 = radius * radius;
    float nx = new_xyz[3 * (size_t)q], ny = new_xyz[3 * (size_t)q + 1], nz = new_xyz[3 * (size_t)q + 2];
    int* o = idx + (size_t)q * nsample;
    int cnt = 0;
    for (int k = 0; k < n; ++k) {
      float x = p[k * 3], y = p[k * 3 + 1], z = p[k * 3 + 2];
      float d2 = (nx - x) * (nx - x) + (ny - y) * (ny - y) + (nz - z) * (nz - z);
      if (d2 < r2) {
        if (cnt == 0) for (int l = 0; l < nsample; ++l) o[l] = k;
        o[cnt] = k;
        ++cnt;
        if (cnt >= nsample) break;
      }
    }
    if (cnt == 0) o[0] = -1;
  }
}

void oracle_group_points(int B, int M, int C, int ns, const float* feat, const int* feat_cnt, const int* idx,
                         const int* idx_cnt, float* out) {
  for (int m = 0; m < M; ++m) {
    int s0 = start_of(feat_cnt, batch_of(idx_cnt, B, m));
    for (int c = 0; c < C; ++c)
      for (int s = 0; s < ns; ++s)
        out[((size_t)m * C + c) * ns + s] = feat[(size_t)(s0 + idx[(size_t)m * ns + s]) * C + c];
  }
}

void oracle_group_points_grad(int B, int M, int C, int N, int ns, const float* grad_out, const int* idx,
                              const int* idx_cnt, const int* feat_cnt, float* grad_feat) {
  double* acc = (double*)calloc((size_t)N * C, sizeof(double));
  for (int m = 0; m < M; ++m) {
    int s0 = start_of(feat_cnt, batch_of(idx_cnt, B, m));
    for (int c = 0; c < C; ++c)
      for (int s = 0; s < ns; ++s)
        acc[(size_t)(s0 + idx[(size_t)m * ns + s]) * C + c] += grad_out[((size_t)m * C + c) * ns + s];
  }
  for (size_t t = 0; t < (size_t)N * C; ++t) grad_feat[t] = (float)acc[t];
  free(acc);
}

/* emulates the block of bs threads + LDS tree of the reference, including its tie behaviour */
void oracle_fps(int B, int n, int m, const float* xyz_all, int* out_all) {
  int pow2 = 0;
  while ((2 << pow2) <= n) ++pow2;
  int bs = 1 << pow2;
  if (bs > 1024) bs = 1024;
  float* temp = (float*)malloc(sizeof(float) * (size_t)n);
  float* dists = (float*)malloc(sizeof(float) * bs);
  int* dists_i = (int*)malloc(sizeof(int) * bs);
  for (int b = 0; b < B; ++b) {
    const float* d = xyz_all + (size_t)b * n * 3;
    int* out = out_all + (size_t)b * m;
    for (int k = 0; k < n; ++k) temp[k] = 1e10f;
    int old = 0;
    if (m > 0) out[0] = 0;
    for (int j = 1; j < m; ++j) {
      float x1 = d[old * 3], y1 = d[old * 3 + 1], z1 = d[old * 3 + 2];
      for (int tid = 0; tid < bs; ++tid) {
        int besti = 0;
        float best = -1;
        for (int k = tid; k < n; k += bs) {
          float x2 = d[k * 3], y2 = d[k * 3 + 1], z2 = d[k * 3 + 2];
          float dd = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
          float d2 = dd < temp[k] ? dd : temp[k];
          temp[k] = d2;
          besti = d2 > best ? k : besti;
          best = d2 > best ? d2 : best;
        }
        dists[tid] = best;
        dists_i[tid] = besti;
      }
      for (int s = bs / 2; s >= 1; s /= 2)
        for (int tid = 0; tid < s; ++tid) {
          float v1 = dists[tid], v2 = dists[tid + s];
          int i1 = dists_i[tid], i2 = dists_i[tid + s];
          dists[tid] = v1 > v2 ? v1 : v2;
          dists_i[tid] = v2 > v1 ? i2 : i1;
        }
      old = dists_i[0];
      out[j] = old;
    }
  }
  free(temp); free(dists); free(dists_i);
}

void oracle_three_nn(int B, int N, const float* unknown, const int* unknown_cnt, const float* known,
                     const int* known_cnt, float* dist2, int* idx) {
  for (int i = 0; i < N; ++i) {
    int b = batch_of(unknown_cnt, B, i);
    int s0 = start_of(known_cnt, b), n = known_cnt[b];
    const float* kp = known + 3 * (size_t)s0;
    float ux = unknown[3 * (size_t)i], uy = unknown[3 * (size_t)i + 1], uz = unknown[3 * (size_t)i + 2];
    double b1 = 1e40, b2 = 1e40, b3 = 1e40;
    int i1 = 0, i2 = 0, i3 = 0;
    for (int k = 0; k < n; ++k) {
      float x = kp[k * 3], y = kp[k * 3 + 1], z = kp[k * 3 + 2];
      float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
      if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; }
      else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = k; }
      else if (d < b3) { b3 = d; i3 = k; }
    }
    dist2[3 * (size_t)i] = (float)b1; dist2[3 * (size_t)i + 1] = (float)b2; dist2[3 * (size_t)i + 2] = (float)b3;
    idx[3 * (size_t)i] = i1 + s0; idx[3 * (size_t)i + 1] = i2 + s0; idx[3 * (size_t)i + 2] = i3 + s0;
  }
}

void oracle_three_interpolate(int N, int C, const float* feat, const int* idx, const float* w, float* out) {
  for (int i = 0; i < N; ++i)
    for (int c = 0; c < C; ++c)
      out[(size_t)i * C + c] = w[3 * (size_t)i] * feat[(size_t)idx[3 * (size_t)i] * C + c] +
                               w[3 * (size_t)i + 1] * feat[(size_t)idx[3 * (size_t)i + 1] * C + c] +
                               w[3 * (size_t)i + 2] * feat[(size_t)idx[3 * (size_t)i + 2] * C + c];
}

void oracle_three_interpolate_grad(int N, int C, int M, const float* grad_out, const int* idx, const float* w,
                                   float* grad_feat) {
  double* acc = (double*)calloc((size_t)M * C, sizeof(double));
  for (int i = 0; i < N; ++i)
    for (int c = 0; c < C; ++c)
      for (int t = 0; t < 3; ++t)
        acc[(size_t)idx[3 * (size_t)i + t] * C + c] += (double)(grad_out[(size_t)i * C + c] * w[3 * (size_t)i + t]);
  for (size_t t = 0; t < (size_t)M * C; ++t) grad_feat[t] = (float)acc[t];
  free(acc);
}

/* one rotation + extent test for both margins: the .cu kernel's 1e-5 (roiaware_pool3d_kernel.cu:23-36) and the CPU
 * twin's 1e-2 (roiaware_pool3d.cpp:121-141, check_pt_in_box3d_cpu). The CPU twin is compiled from the reference's own
 * source into oracle/_ref (build_ref.sh) and tests/test_oracle_pointnet2.py pins oracle_points_in_boxes_cpu to it
 * bit-exactly, which pins this rotation arithmetic. */
static int pt_in_box_margin(const float* pt, const float* box, float* lx, float* ly, const float MARGIN) {
  float x = pt[0], y = pt[1], z = pt[2];
  float cx = box[0], cy = box[1], cz = box[2], dx = box[3], dy = box[4], dz = box[5], rz = box[6];
  if (fabsf(z - cz) > dz / 2.0) return 0;
  float cosa = cosf(-rz), sina = sinf(-rz);
  float sx = x - cx, sy = y - cy;
  *lx = sx * cosa + sy * (-sina);
  *ly = sx * sina + sy * cosa;
  return (fabs(*lx) < dx / 2.0 + MARGIN) & (fabs(*ly) < dy / 2.0 + MARGIN);
}
static int pt_in_box(const float* pt, const float* box, float* lx, float* ly) {
  return pt_in_box_margin(pt, box, lx, ly, 1e-5f);
}

/* points_in_boxes_cpu (roiaware_pool3d.cpp:144-167): (N,P) membership matrix, MARGIN = 1e-2 */
void oracle_points_in_boxes_cpu(int N, int P, const float* boxes, const float* pts, int* out) {
  float lx = 0, ly = 0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < P; ++j)
      out[(size_t)i * P + j] = pt_in_box_margin(pts + 3 * (size_t)j, boxes + 7 * (size_t)i, &lx, &ly, 1e-2f);
}

void oracle_points_in_boxes(int B, int T, int M, const float* boxes, const float* pts, int* out) {
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < M; ++i) {
      const float* p = pts + ((size_t)b * M + i) * 3;
      int found = -1;
      float lx = 0, ly = 0;
      for (int k = 0; k < T; ++k)
        if (pt_in_box(p, boxes + ((size_t)b * T + k) * 7, &lx, &ly)) { found = k; break; }
      out[(size_t)b * M + i] = found;
    }
}

/* forward of the RoI-aware pool: fills pts_idx_of_voxels (N,ox,oy,oz,mp), pooled (N,ox,oy,oz,C), argmax */
void oracle_roiaware_pool(int N, int P, int C, int mp, int ox, int oy, int oz, const float* rois, const float* pts,
                          const float* feat, int* argmax, int* pts_idx, float* pooled, int method) {
  memset(pts_idx, 0, sizeof(int) * (size_t)N * ox * oy * oz * mp);
  memset(pooled, 0, sizeof(float) * (size_t)N * ox * oy * oz * C);
  memset(argmax, 0, sizeof(int) * (size_t)N * ox * oy * oz * C);
  for (int bi = 0; bi < N; ++bi) {
    const float* r = rois + 7 * (size_t)bi;
    int* vox = pts_idx + (size_t)bi * ox * oy * oz * mp;
    for (int k = 0; k < P; ++k) {
      float lx = 0, ly = 0;
      if (!pt_in_box(pts + 3 * (size_t)k, r, &lx, &ly)) continue;
      float lz = pts[3 * (size_t)k + 2] - r[2];
      float dx = r[3], dy = r[4], dz = r[5];
      float xr = dx / ox, yr = dy / oy, zr = dz / oz;
      unsigned xi = (unsigned)(int)((lx + dx / 2) / xr), yi = (unsigned)(int)((ly + dy / 2) / yr),
               zi = (unsigned)(int)((lz + dz / 2) / zr);
      if (xi > (unsigned)(ox - 1)) xi = ox - 1;
      if (yi > (unsigned)(oy - 1)) yi = oy - 1;
      if (zi > (unsigned)(oz - 1)) zi = oz - 1;
      int* v = vox + ((size_t)(xi * oy + yi) * oz + zi) * mp;
      if (v[0] < mp - 1) { v[v[0] + 1] = k; v[0]++; }
    }
    for (int cell = 0; cell < ox * oy * oz; ++cell) {
      const int* v = vox + (size_t)cell * mp;
      for (int c = 0; c < C; ++c) {
        size_t o = ((size_t)bi * ox * oy * oz + cell) * C + c;
        if (method == 0) {
          int am = -1;
          float mx = -INFINITY;
          for (int k = 1; k <= v[0]; ++k) {
            float f = feat[(size_t)v[k] * C + c];
            if (f > mx) { mx = f; am = v[k]; }
          }
          if (am != -1) pooled[o] = mx;
          argmax[o] = am;
        } else {
          float s = 0;
          for (int k = 1; k <= v[0]; ++k) s += feat[(size_t)v[k] * C + c];
          if (v[0] > 0) pooled[o] = s / v[0];
        }
      }
    }
  }
}
