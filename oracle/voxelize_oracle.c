/*
 * ORACLE — test infrastructure only (never linked, imported or called by the product path).
 * CPU restatement of the voxel generator the reference calls through
 * pcdet/datasets/processor/data_processor.py:44-60 (VoxelGeneratorWrapper.generate ->
 * spconv.utils.Point2VoxelCPU3d.point_to_voxel).  The algorithm lives in the third-party wheel
 * spconv-cu113 v2.1.21 (README.md:54), absent from /root/reference: this follows its published
 * sequential algorithm (first-occurrence voxel ids, first max_points points kept, new voxels dropped
 * once max_voxels exist).  PARITY UNPINNED: the reference holds no test or golden vector for it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* One frame. Returns number of voxels. coords are [z,y,x] like the reference (data_processor.py:56). */
int oracle_voxelize_frame(const float* pts, int n, int C, const float* range_min, const float* vsize,
                          const int* grid_xyz, int max_voxels, int max_points,
                          float* voxels /* (max_voxels,max_points,C) zeroed by us */,
                          int* coords /* (max_voxels,3) */, int* num_points /* (max_voxels) */) {
  const int64_t gx = grid_xyz[0], gy = grid_xyz[1], gz = grid_xyz[2];
  int* grid = (int*)malloc(sizeof(int) * (size_t)(gx * gy * gz));
  if (!grid) return -1;
  memset(grid, 0xff, sizeof(int) * (size_t)(gx * gy * gz));
  memset(voxels, 0, sizeof(float) * (size_t)max_voxels * max_points * C);
  memset(num_points, 0, sizeof(int) * (size_t)max_voxels);
  int voxel_num = 0;
  for (int i = 0; i < n; ++i) {
    const float* p = pts + (size_t)i * C;
    int c[3];
    int failed = 0;
    for (int j = 0; j < 3; ++j) {
      float q = (p[j] - range_min[j]) / vsize[j];
      if (!(q == q)) { failed = 1; break; }
      float f = floorf(q);
      if (f < 0.0f || f >= (float)grid_xyz[j]) { failed = 1; break; }
      c[j] = (int)f;
    }
    if (failed) continue;
    int64_t lin = ((int64_t)c[2] * gy + c[1]) * gx + c[0];
    int vid = grid[lin];
    if (vid == -1) {
      if (voxel_num >= max_voxels) continue;
      vid = voxel_num++;
      grid[lin] = vid;
      coords[vid * 3 + 0] = c[2];
      coords[vid * 3 + 1] = c[1];
      coords[vid * 3 + 2] = c[0];
    }
    int k = num_points[vid];
    if (k < max_points) {
      memcpy(voxels + ((size_t)vid * max_points + k) * C, p, sizeof(float) * C);
      num_points[vid] = k + 1;
    }
  }
  free(grid);
  return voxel_num;
}

/* MeanVFE (pcdet/models/backbones_3d/vfe/mean_vfe.py:14-31): sum over the point axis / clamp_min(count,1) */
void oracle_mean_vfe(const float* voxels, const int* num_points, int M, int max_points, int C, float* out) {
  for (int m = 0; m < M; ++m) {
    float k = (float)(num_points[m] < 1 ? 1 : num_points[m]);
    for (int c = 0; c < C; ++c) {
      float s = 0.f;
      for (int t = 0; t < max_points; ++t) s += voxels[((size_t)m * max_points + t) * C + c];
      out[(size_t)m * C + c] = s / k;
    }
  }
}
