/*
 * libcrbhip.so — C-ABI of the MI355X (gfx950) hot path of CRB-active-3Ddet.
 *
 * Every entry point: plain device pointers + sizes, caller-owned outputs AND workspace, launches on the
 * hipStream_t passed as `void* stream`, never synchronises unless stated, returns 0 or a negative
 * CRB_ERR_* code (never exit()). No torch / pybind types cross this boundary.
 *
 * Each declaration cites the reference interface (file:line under the upstream tree) it replaces.
 * The Python host side (crb-active-3ddet_amd/crbhip/_lib.py) binds these with ctypes; INTEGRATION.md
 * shows the stub a pcdet maintainer would add.
 */
#ifndef CRB_HIP_H
#define CRB_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRB_OK 0
#define CRB_ERR_ARG (-1)
#define CRB_ERR_WORKSPACE (-2)
#define CRB_ERR_LAUNCH (-3)
#define CRB_ERR_UNSUPPORTED (-4)

/* ABI version of this header; bumped on any signature change. */
int crb_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * a1/a2/a3  VoxelGenerator (+ collate + MeanVFE)
 * replaces: spconv.utils.Point2VoxelCPU3d.point_to_voxel as called from
 *           pcdet/datasets/processor/data_processor.py:44-60 (VoxelGeneratorWrapper.generate),
 *           the voxel concat / batch-index pad of pcdet/datasets/dataset.py:160-229 (collate_batch),
 *           and pcdet/models/backbones_3d/vfe/mean_vfe.py:14-31 (MeanVFE.forward) when mean_features != NULL.
 *
 * points          (n_points, num_features) f32, frames concatenated, xyz first
 * frame_offsets   (B+1) i32 device, frame b owns points [off[b], off[b+1])
 * range_min_xyz / voxel_size_xyz / grid_xyz : HOST arrays of 3
 * outputs sized for cap = min(B*max_voxels, n_points) rows:
 *   voxels (cap, max_points, num_features) f32 or NULL; coords (cap,4) i32 [b,z,y,x];
 *   num_points (cap) i32; mean_features (cap, num_features) f32 or NULL;
 *   num_voxels_out (B+1) i32 device: per-frame kept voxel count, [B] = total rows written.
 * Rows beyond the total are left untouched.
 * ---------------------------------------------------------------------------------------------- */
int64_t crb_voxelize_workspace_bytes(int64_t n_points, int B, int max_voxels, int max_points);
int crb_voxelize(const float* points, int64_t n_points, int num_features,
                 const int32_t* frame_offsets, int B,
                 const float* range_min_xyz, const float* voxel_size_xyz, const int32_t* grid_xyz,
                 int max_voxels, int max_points,
                 float* voxels, int32_t* coords, int32_t* num_points, float* mean_features,
                 int32_t* num_voxels_out,
                 void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a4  Sparse 3D convolution: rulebooks + gather-GEMM fwd / dgrad / wgrad
 * replaces: spconv.pytorch.SubMConv3d / SparseConv3d / SparseConvTensor (third-party spconv-cu113 v2.1.21,
 *           not vendored) as used by pcdet/models/backbones_3d/spconv_backbone.py:8-27,77-117,141-157 and
 *           pcdet/utils/spconv_utils.py:3-34.
 *
 * coords (N,4) i32 [b,z,y,x]; shape_dhw = spatial shape [D,H,W] (HOST int32[3]); ksize/stride/padding HOST int32[3].
 * Kernel offset index o = (kz*KH + ky)*KW + kx; weights are (K, Cin, Cout) f32 contiguous.
 * ---------------------------------------------------------------------------------------------- */
int64_t crb_hash_capacity_for(int64_t n);   /* power of two >= 2n */
/* site -> row hash for a coordinate set in ANY row order (the voxelizer's output): hkeys (capacity) i64, hvals (capacity) i32.
 * Slots are grouped by 8 consecutive x-sites (slot = group(site / 8) * 8 + site % 8, whole groups probe linearly), so the
 * three x-neighbours of a kernel row usually share one 64-byte line of keys. Duplicate coordinates: the smallest row wins.
 * Consumed by crb_subm_rows; sets that come out of crb_spconv_chain_emit need no hash (rank tables). */
int crb_sparse_hash_build(const int32_t* coords, int64_t n, const int32_t* shape_dhw,
                          int64_t* hkeys, int32_t* hvals, int64_t capacity, void* stream);
/* Y (n_out,cout) = sum_o X[nbr[:,o]] @ W[o]; also used for dgrad with the transposed table / weights.
 * Optional row permutation: when perm != NULL, `nbr` holds the table rows in permuted order (nbr_sorted[s] =
 * nbr[perm[s]]) and the result of sorted row s is written to Y[perm[s]]. Sorting rows by their neighbour bit-mask
 * (crb_nbr_masks -> sort -> crb_nbr_permute) lets a wave skip every kernel offset none of its 16 rows uses. */
int crb_sparse_conv_supported(int cin, int cout);
int crb_nbr_masks(const int32_t* nbr, int64_t n, int K, int32_t* mask, void* stream);
/* stable sort of the rows of every chunk of crb_mask_sort_chunk_rows() consecutive rows by mask, DESCENDING (rows with the
 * most neighbours first) -> perm (n) */
int crb_mask_sort_chunk_rows(void);
/* sort key of crb_mask_sort_chunks: the mask with its bits re-ranked rarest offset first by frequency inside the chunk
 * (A/B alternatives: crb_hip_measure.h). The gather-GEMM's results do not depend on the row order; the MFMA tile fill does. */
int crb_mask_sort_chunks(const int32_t* mask, int64_t n, int32_t* perm, void* stream);
/* the same order for chunks of chunk_rows rows (a multiple of 1024; chunks too large for one workgroup's LDS): keys ranked
 * per chunk, then one stable device radix sort (hipCUB) of (chunk, key, row) */
int64_t crb_mask_sort_rows_workspace_bytes(int64_t n);
int crb_mask_sort_rows(const int32_t* mask, int64_t n, int chunk_rows, int32_t* perm, void* workspace,
                       int64_t workspace_bytes, void* stream);
/* perm_out = perm_in with the full 64-row tiles of each of crb_tile_lpt_ranges() contiguous tile ranges (one per XCD, the
 * split the gather-GEMM's workgroup->tile map uses) re-ordered heaviest first; weight = kernel offsets present in any row
 * of the tile = phases its workgroup will run. A launch then ends on light tiles instead of idling behind 27-phase ones,
 * and a range's gathers stay in one XCD's L2. mask as written by crb_nbr_masks (indexed by original row). The order among
 * equal weights is arbitrary; the trailing partial tile stays last. */
int crb_tile_lpt_ranges(void);
int64_t crb_tile_lpt_workspace_bytes(int64_t n);
int crb_tile_lpt_perm(const int32_t* mask, const int32_t* perm_in, int64_t n, int32_t* perm_out, void* workspace,
                      int64_t workspace_bytes, void* stream);
int crb_nbr_permute(const int32_t* nbr, const int32_t* perm, int64_t n, int K, int32_t* nbr_sorted, void* stream);
/* Compact neighbour table of the gather-GEMM: cmask (n) u32 = kernel offsets present in row i of the KERNEL order (row
 * perm[i] of nbr; perm NULL = identity), cbase (n+1) i32 = exclusive prefix of the masks' popcounts, packed (>= cbase[n],
 * caller allocates n*K) = the present neighbour indices row after row, offsets ascending. 8 + 4 P/n bytes per row instead
 * of 4 K. crb_sparse_conv_forward_compact is crb_sparse_conv_forward on that table (bit-identical results); shapes for
 * which crb_sparse_conv_compact_supported() is 0 keep the (n,K) table. Supported: the phase kernel's shapes (Cin, Cout multiples of
 * 16, Cin <= 64) and, since round 4, the low-channel forward instance (Cin, Cout) = (4, 16) of the KITTI input layer (a wave owns a
 * 16-row tile end to end). */
int crb_sparse_conv_compact_supported(int cin, int cout);
int64_t crb_nbr_compact_workspace_bytes(int64_t n);
int crb_nbr_compact(const int32_t* nbr, const int32_t* perm, int64_t n, int K, uint32_t* cmask, int32_t* cbase,
                    int32_t* packed, void* workspace, int64_t workspace_bytes, void* stream);
/* tile_order (ceil(n_out/64)) or NULL: position -> 64-row tile of the kernel order, the heaviest-first dispatch order inside
 * each XCD's contiguous range of tiles (crb_tables_finish); results do not depend on it. */
int crb_sparse_conv_forward_compact(const float* X, const float* W, const uint32_t* cmask, const int32_t* cbase,
                                    const int32_t* packed, const int32_t* perm, const int32_t* tile_order, float* Y,
                                    int64_t n_out, int K, int cin, int cout, void* stream);

/* ---- the whole backbone's tables in ~26 launches and ONE host read-back (csrc/rulebook_plan.hip) --------------------------
 * replaces: the indice-pair generation of every spconv.pytorch.SubMConv3d / SparseConv3d of
 *   pcdet/models/backbones_3d/spconv_backbone.py:77-117 (VoxelBackBone8x: 4 SubM keys + 4 strided keys), which spconv builds
 *   layer by layer inside forward with one synchronisation per strided layer.
 * chain: output sets of a chain of strided convs (level l+1 = conv l+1 applied to level l's set). geoms = n_levels x 9 ints
 *   {k,s,p} (d,h,w), out_shapes = n_levels x 3; the bitmaps of all levels live in one caller-owned buffer, level l at word
 *   word_off[l] (n_levels+1 entries, multiples of 2048, word_off[l+1]-word_off[l] >= crb_spconv_padded_words of level l);
 *   tile_sums_all holds one int per 2048-word tile. crb_spconv_chain_mark zero-fills and marks every level and writes the
 *   per-level site counts to counts_dev (device; the caller reads them back ONCE and sizes every later buffer).
 *   crb_spconv_chain_emit: per level the coordinates (ascending (b,z,y,x), n_out[l] rows, host array of device pointers) and
 *   the rank table: 8 bytes per bitmap word {bits, rows before the word} so that site -> row is ONE load. */
int64_t crb_spconv_padded_words(int B, const int32_t* out_shape_dhw);
int crb_spconv_chain_mark(const int32_t* coords, int64_t n, int B, const int32_t* in_shape_dhw, int n_levels,
                          const int32_t* geoms, const int32_t* out_shapes, const int64_t* word_off, uint32_t* bitmap_all,
                          int32_t* tile_sums_all, int32_t* counts_dev, void* stream);
/* the same with the input row count still on the device: coords holds n_cap rows of which the first *n_dev are valid (the voxel
 * generator's coordinate buffer and its total, crb_voxelize's counts[B]). The chain is marked and counted before the host knows
 * the voxel count; ONE read-back then returns it together with the level sizes (the reference path synchronises twice here:
 * Point2VoxelCPU3d returns host arrays, spconv's indice generation returns host-side counts per strided layer). */
int crb_spconv_chain_mark_lazy(const int32_t* coords, int64_t n_cap, const int32_t* n_dev, int B, const int32_t* in_shape_dhw,
                               int n_levels, const int32_t* geoms, const int32_t* out_shapes, const int64_t* word_off,
                               uint32_t* bitmap_all, int32_t* tile_sums_all, int32_t* counts_dev, void* stream);
int crb_spconv_chain_emit(int B, int n_levels, const int32_t* out_shapes, const int64_t* word_off,
                          const uint32_t* bitmap_all, const int32_t* tile_sums_all, void* rank_all,
                          int32_t* const* out_coords, const int64_t* n_out, void* stream);
/* rows: ONE kernel per table writes the neighbour rows (n,K), mask (n) u32 = offsets present per row and hist
 * (ceil(n / crb_table_chunk_rows()), 32) i32 += per-chunk count of every offset (caller zero-fills hist).
 * crb_subm_rows: SubM table of a site set; sites are looked up in `rank` (rank table of THIS set from crb_spconv_chain_emit:
 *   rows are bitmap ranks) or, rank == NULL, in the site hash of crb_sparse_hash_build (any row order).
 * crb_spconv_rows: strided conv from the input rows `coords` to the set whose rank table is rank_out: nbr_t (n,K) = output row
 *   per (input row, offset) with its mask / hist, and nbr (n_out,K) filled by the unique writer of every entry (the launch
 *   first fills it with -1). No existence test is needed (every site an input reaches is an output).
 * crb_table_masks: mask + hist of an existing (n,K) table. */
int crb_table_chunk_rows(void);
int crb_subm_rows(const int32_t* coords, int64_t n, const int32_t* shape_dhw, const int32_t* ksize, const int64_t* hkeys,
                  const int32_t* hvals, int64_t capacity, const void* rank, int32_t* nbr, uint32_t* mask, int32_t* hist,
                  void* stream);
int crb_spconv_rows(const int32_t* coords, int64_t n, const int32_t* ksize, const int32_t* stride, const int32_t* padding,
                    const int32_t* out_shape_dhw, const void* rank_out, int64_t n_out, int32_t* nbr, int32_t* nbr_t,
                    uint32_t* mask_t, int32_t* hist_t, void* stream);
int crb_table_masks(const int32_t* nbr, int64_t n, int K, uint32_t* mask, int32_t* hist, void* stream);
/* finish: everything the kernels read, for any number of tables, in THREE launches per 16 tables (sort pass: one
 * workgroup per 4096-row chunk, register/shuffle bitonic network; fill pass: one workgroup per 256 rows; tile order).
 * per table (host array of CrbTablePlan; all pointers device, caller-owned):
 *   in : nbr (n,K), mask (n), hist (chunks,32) as written by the rows kernels
 *   out: perm (n) kernel order of the rows (every 4096-row chunk sorted by mask, rarest offset first, descending, stable);
 *        cmask (n) / cbase (n+1) / packed (>= P, caller allocates n*K) the compact table of crb_nbr_compact in that order;
 *        tile_weight / tile_order (ceil(n/64)) offsets present per 64-row tile and the heaviest-first order of the tiles
 *        inside each of the 8 XCD ranges (stable: deterministic);
 *        pair_in / pair_out (>= P) / pair_start (K+1) + workspace pair_unit_base, or all four NULL: the wgrad pair lists of
 *        wgrad kernels: pair_in / pair_out sorted by (offset, output row), pair_start[o] = first pair of offset o, [K] = P. */
typedef struct CrbTablePlan {
  const int32_t* nbr;
  const uint32_t* mask;
  const int32_t* hist;
  int32_t* perm;
  uint32_t* cmask;
  int32_t* cbase;
  int32_t* packed;
  int32_t* tile_weight;
  int32_t* tile_order;
  int32_t* pair_in;
  int32_t* pair_out;
  int32_t* pair_start;
  int32_t* pair_unit_base;      /* workspace, ceil(n/64) x 32 ints (NULL with the pair lists): first pair of every (64-row unit, offset) */
  int64_t n;
  int32_t K;
  int32_t reserved;
} CrbTablePlan;
int crb_tables_finish(const CrbTablePlan* tables, int n_tables, void* stream);

/* Inference epilogue in the gather-GEMM: y = relu(gamma * ((conv + bias - running_mean) * rsqrt(running_var + eps)) + beta),
 * i.e. the (bias,) nn.BatchNorm1d in eval mode and nn.ReLU that follow a sparse conv in post_act_block
 * (pcdet/models/backbones_3d/spconv_backbone.py:8-31), evaluated on the accumulator before the store instead of in two more
 * passes over the rows; same arithmetic order as crb_bn_relu_apply. bias may be NULL; relu 0/1. Shapes of
 * crb_sparse_conv_compact_supported only. */
int crb_sparse_conv_forward_compact_bn(const float* X, const float* W, const uint32_t* cmask, const int32_t* cbase,
                                       const int32_t* packed, const int32_t* perm, const int32_t* tile_order, float* Y,
                                       int64_t n_out, int K, int cin, int cout, const float* bias, const float* gamma, const float* beta,
                                       const float* running_mean, const float* running_var, float eps, int relu,
                                       void* stream);

/* the weight layouts of n <= 32 layers in one launch: w[j] = a spconv layer's parameter (Cout_j, K_j, Cin_j) contiguous (spconv 2.x
 * checkpoint layout, spconv/pytorch/conv.py) -> w_kio[j] (K, Cin, Cout) = the forward operand W[o] of crb_sparse_conv_forward*, and
 * w_dgrad[j] (K, Cout, Cin) (NULL = not wanted) = the input gradient's operand: W[o]^T, at K-1-o when flip[j] (submanifold layers).
 * Replaces a permuted copy per layer forward and a flip + transposed copy per layer backward. Host arrays of pointers / ints. */
int crb_sparse_weights_multi(int n, const float* const* w, const int32_t* K, const int32_t* cin, const int32_t* cout,
                             const int32_t* flip, float* const* w_kio, float* const* w_dgrad, void* stream);
int crb_sparse_conv_forward(const float* X, const float* W, const int32_t* nbr, const int32_t* perm, float* Y,
                            int64_t n_out, int K, int cin, int cout, void* stream);
/* dW (K,cin,cout) = sum over pairs X[pin]^T dY[pout] */
int crb_sparse_conv_wgrad_splits(void);
int crb_sparse_conv_wgrad_occupancy(int cin, int cout); /* measurement helper: resident workgroups per CU of the v2 wgrad instance */
int64_t crb_sparse_conv_wgrad_workspace_bytes(int K, int cin, int cout);
/* Windowed wgrad (tiles of <= 4 blocks of 32x32): work is cut by WINDOWS OF OUTPUT ROWS instead of per-offset pair ranges —
 * XCD x owns the x-th eighth of the rows, all kernel offsets' workgroups of an XCD walk the same window at the same time,
 * so gathered rows are served by the XCD's L2 instead of being fetched once per offset. bounds (K, crb_wgrad_num_windows()+1)
 * = first pair of every window per offset (crb_wgrad_window_bounds, once per rulebook; pair_out ascending inside an offset).
 * Same deterministic partial + fixed-order reduction scheme as crb_sparse_conv_wgrad, same result up to f32 summation
 * order. */
int crb_wgrad_num_windows(void);
int crb_wgrad_window_bounds(const int32_t* pair_out, const int32_t* pair_start, int K, int64_t n_out, int32_t* bounds,
                            void* stream);
int crb_sparse_conv_wgrad_windowed_supported(int cin, int cout);
int64_t crb_sparse_conv_wgrad_windowed_workspace_bytes(int K, int cin, int cout);
int crb_sparse_conv_wgrad_windowed(const float* X, const float* dY, const int32_t* pair_in, const int32_t* pair_out,
                                   const int32_t* pair_start, const int32_t* bounds, float* dW, int K, int cin, int cout,
                                   void* workspace, int64_t workspace_bytes, void* stream);
int crb_sparse_conv_wgrad(const float* X, const float* dY, const int32_t* pair_in, const int32_t* pair_out,
                          const int32_t* pair_start, float* dW, int K, int cin, int cout,
                          void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a6  SparseConvTensor.dense() and its backward
 * replaces: spconv SparseConvTensor.dense() as used by
 *           pcdet/models/backbones_2d/map_to_bev/height_compression.py:20-24.
 * out (B,C,D,H,W) f32. zero_fill != 0 clears `out` first.
 * ---------------------------------------------------------------------------------------------- */
int crb_sparse_to_dense(const float* feat, const int32_t* coords, float* out, int64_t n, int B, int C,
                        int D, int H, int W, int zero_fill, void* stream);
int crb_dense_to_sparse(const float* dense, const int32_t* coords, float* feat, int64_t n, int B, int C,
                        int D, int H, int W, void* stream);
/* same, BEV channels-last memory: out is (B, H, W, C*D) with channel index c*D + z — the layout of a
 * torch (B, C*D, H, W) tensor in channels_last format, i.e. HeightCompression's view without NCHW<->NHWC transposes */
int crb_sparse_to_dense_nhwc(const float* feat, const int32_t* coords, float* out, int64_t n, int B, int C,
                             int D, int H, int W, int zero_fill, void* stream);
int crb_dense_to_sparse_nhwc(const float* dense, const int32_t* coords, float* feat, int64_t n, int B, int C,
                             int D, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a13/a14  rotated BEV overlap / IoU / 3-D IoU and NMS
 * replaces: the pybind surface of pcdet/ops/iou3d_nms/src/iou3d_nms_api.cpp:11-17
 *   boxes_overlap_bev_gpu (iou3d_nms.cpp:46-66)  -> crb_boxes_pairwise(mode 0)
 *   boxes_iou_bev_gpu     (iou3d_nms.cpp:68-88)  -> crb_boxes_pairwise(mode 1)
 *   boxes_iou3d_gpu       (iou3d_nms_utils.py:48-81, fused) -> crb_boxes_pairwise(mode 2)
 *   nms_gpu / nms_normal_gpu (iou3d_nms.cpp:90-185: device mask + D2H + host greedy scan) -> crb_nms_batched
 * boxes are (n,7) f32 [x,y,z,dx,dy,dz,heading]; out (na,nb) f32.
 * crb_nms_batched: boxes_sorted (B,nmax,7) already in descending score order, counts (B) i32 device or NULL (= nmax
 * everywhere); keep (B,max_keep) i32 indices into the sorted order, -1 padded; num_keep (B) i32. Greedy scan runs on
 * the device; no synchronisation. workspace >= crb_nms_workspace_bytes(B,nmax).
 * ---------------------------------------------------------------------------------------------- */
int crb_boxes_pairwise(const float* boxes_a, int64_t na, const float* boxes_b, int64_t nb, float* out,
                       int mode, void* stream);
int64_t crb_nms_workspace_bytes(int B, int64_t nmax);
int crb_nms_batched(const float* boxes_sorted, const int32_t* counts, int B, int64_t nmax, float thresh,
                    int rotated, int max_keep, int32_t* keep, int32_t* num_keep, void* workspace,
                    int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a16-a18  PointNet++ stack ops
 * replaces: the pybind surface of pcdet/ops/pointnet2/pointnet2_stack/src/pointnet2_api.cpp:12-31
 *   ball_query_wrapper (ball_query.cpp:31-47), group_points_wrapper / group_points_grad_wrapper
 *   (group_points.cpp), farthest_point_sampling_wrapper (sampling.cpp), three_nn_wrapper,
 *   three_interpolate_wrapper / three_interpolate_grad_wrapper (interpolate.cpp).
 * Stacked layout: xyz (N1+N2+..,3) f32 + *_batch_cnt (B) i32 on the device; idx int32.
 * ball query: idx (M,nsample): hits in scan order, padded with the first hit; empty ball: idx[m][0] = -1, rest 0.
 * grad kernels accumulate with atomics into PRE-ZEROED outputs.
 * ---------------------------------------------------------------------------------------------- */
int crb_ball_query_stack(int B, int64_t M, float radius, int nsample, const float* new_xyz,
                         const int32_t* new_xyz_batch_cnt, const float* xyz, const int32_t* xyz_batch_cnt,
                         int32_t* idx, void* stream);
/* two radii, same centres, one scan. idx_a (M,nsample_a) / idx_b (M,nsample_b) already in the grouping kernels' form (an
 * empty ball is all zeros, flagged in empty_a / empty_b (M) u8): replaces two ball_query_wrapper calls plus the Python
 * fix-up `empty = idx[:,0]==-1; idx[empty]=0` (pointnet2_utils.py:31-38) */
int crb_ball_query2_stack(int B, int64_t M, float radius_a, int nsample_a, float radius_b, int nsample_b,
                          const float* new_xyz, const int32_t* new_xyz_batch_cnt, const float* xyz,
                          const int32_t* xyz_batch_cnt, int32_t* idx_a, int32_t* idx_b, uint8_t* empty_a,
                          uint8_t* empty_b, void* stream);
/* crb_ball_query2_stack on a per-call cell grid (round 6): the call's source points are counting-sorted into a hashed grid of cells
 * of edge 1.001 radius_b (histogram, exclusive scan, cursor fill), a query tests the points of the 27 cells around its own and
 * keeps the nsample smallest indices of each radius in ascending order - the lists of crb_ball_query2_stack, index for index
 * (replaces the same ball_query_wrapper calls, pointnet2_utils.py:31-38 / ball_query_gpu.cu:47-66: hits in scan order, first
 * nsample, padded with the first hit). radius_a <= radius_b. n_total = rows of xyz. workspace: crb_ball_query2_grid_workspace_bytes. */
int64_t crb_ball_query2_grid_workspace_bytes(int64_t n_total);
int crb_ball_query2_grid_stack(int B, int64_t M, float radius_a, int nsample_a, float radius_b, int nsample_b,
                               const float* new_xyz, const int32_t* new_xyz_batch_cnt, const float* xyz,
                               const int32_t* xyz_batch_cnt, int64_t n_total, int32_t* idx_a, int32_t* idx_b, uint8_t* empty_a,
                               uint8_t* empty_b, void* workspace, int64_t workspace_bytes, void* stream);
/* crb_ball_query2_stack for queries that come in spatially compact groups of `group` consecutive rows inside one frame
 * (every new_xyz_batch_cnt[b] a multiple of `group` for the fast path; a group that straddles two frames is still answered
 * correctly, by a per-query scan; the 216 grid points of one RoI in PVRCNNHead.roi_grid_pool,
 * pvrcnn_head.py:97-132, feeding pointnet2_utils.py:31-38): one workgroup per group prefilters the frame's points by the
 * group's bounding sphere, same index lists as crb_ball_query2_stack. */
int crb_ball_query2_grouped_stack(int B, int64_t M, int group, float radius_a, int nsample_a, float radius_b, int nsample_b,
                                  const float* new_xyz, const int32_t* new_xyz_batch_cnt, const float* xyz,
                                  const int32_t* xyz_batch_cnt, int32_t* idx_a, int32_t* idx_b, uint8_t* empty_a,
                                  uint8_t* empty_b, void* stream);
int crb_group_points_stack(int B, int64_t M, int C, int nsample, const float* features,
                           const int32_t* features_batch_cnt, const int32_t* idx,
                           const int32_t* idx_batch_cnt, float* out, void* stream);
int crb_group_points_grad_stack(int B, int64_t M, int C, int nsample, const float* grad_out,
                                const int32_t* idx, const int32_t* idx_batch_cnt,
                                const int32_t* features_batch_cnt, float* grad_features, void* stream);
/* fused QueryAndGroup (pointnet2_utils.py:107-155) in the layout the shared MLP consumes: out (3+C, M, nsample) =
 * [xyz[nbr]-new_xyz ; features[nbr]] per (query, sample), zero for empty balls (empty_mask (M) u8). idx as returned by
 * crb_ball_query_stack after the empty fix-up. grad: features only, atomics into a pre-zeroed (N,C) buffer. */
int crb_query_group_stack(int B, int64_t M, int C, int nsample, const float* xyz,
                          const int32_t* xyz_batch_cnt, const float* features, const float* new_xyz,
                          const int32_t* new_xyz_batch_cnt, const int32_t* idx, const uint8_t* empty_mask,
                          float* out, void* stream);
int crb_query_group_grad_stack(int B, int64_t M, int C, int nsample, const int32_t* xyz_batch_cnt,
                               const int32_t* new_xyz_batch_cnt, const int32_t* idx,
                               const uint8_t* empty_mask, const float* grad_out,
                               float* grad_features, void* stream);
/* the same grouping in row-major layout, out (M*nsample, 3+C): the training path runs the shared MLP as GEMMs + fused
 * BatchNorm/ReLU row kernels on it. The grad kernel pre-sums pairs that share a source row inside a workgroup (ball-query
 * padding repeats the first hit) before its atomics. */
int crb_query_group_rows_stack(int B, int64_t M, int C, int nsample, const float* xyz,
                               const int32_t* xyz_batch_cnt, const float* features, const float* new_xyz,
                               const int32_t* new_xyz_batch_cnt, const int32_t* idx, const uint8_t* empty_mask,
                               float* out, void* stream);
int crb_query_group_rows_grad_stack(int B, int64_t M, int C, int nsample, const int32_t* xyz_batch_cnt,
                                    const int32_t* new_xyz_batch_cnt, const int32_t* idx,
                                    const uint8_t* empty_mask, const float* grad_out, float* grad_features,
                                    void* stream);
/* training path, first shared-MLP layer without the grouped matrix: for a bias-free 1x1 conv on QueryAndGroup's output
 * (pointnet2_utils.py:107-155 feeding pointnet2_modules.py:94-97), y[p] = W1x (xyz_j - c_i) + P[j] with P = features @ W1f^T
 * (N,H) computed by the caller. out (M*nsample, H) row-major, rel (M*nsample, 3) = xyz_j - c_i (0 for empty balls, whose
 * rows of out are 0 like the reference's zeroed groups). The grad entry accumulates grad_P (N,H, pre-zeroed) and writes
 * part (crb_group_affine_rows_grad_blocks(M,nsample), 3, H): per-slab pieces of dW1x, summed by the caller. */
int crb_group_affine_rows_stack(int B, int64_t M, int H, int nsample, const float* xyz, const int32_t* xyz_batch_cnt,
                                const float* P, const float* new_xyz, const int32_t* new_xyz_batch_cnt,
                                const int32_t* idx, const uint8_t* empty_mask, const float* W1x, float* out, float* rel,
                                void* stream);
/* forward that also writes stat (crb_group_affine_rows_grad_blocks(M,nsample), 2, H): the column sums of out and out^2 of every
 * 64-row slab — the statistics pass of the BatchNorm that follows (crb_bn_relu_forward_partials) then reads those instead of
 * the (M*nsample, H) rows. H in {16, 32, 64, 128}. */
int crb_group_affine_rows_stats_stack(int B, int64_t M, int H, int nsample, const float* xyz, const int32_t* xyz_batch_cnt,
                                      const float* P, const float* new_xyz, const int32_t* new_xyz_batch_cnt,
                                      const int32_t* idx, const uint8_t* empty_mask, const float* W1x, float* out, float* rel,
                                      float* stat, void* stream);
int64_t crb_group_affine_rows_grad_blocks(int64_t M, int nsample);
/* the grad entry with the BatchNorm(+ReLU) backward of the layer's output folded into its slab loads (the BatchNorm2d + ReLU
 * that follow the first conv of a shared MLP, pointnet2_modules.py:94-99): grad_z = gradient w.r.t. relu(batchnorm(y)),
 * y = the forward's `out`, mean / invstd the batch statistics, dbeta / dgamma the reduced gradients (crb_bn_relu_backward with
 * dx = NULL): dy = gamma invstd (d - dbeta/n - xhat dgamma/n), d = grad_z [z > 0], n = M*nsample, is formed in registers — the
 * (M*nsample, H) gradient of y is neither written nor read. */
int crb_group_affine_rows_grad_bn_stack(int B, int64_t M, int H, int nsample, const int32_t* xyz_batch_cnt,
                                        const int32_t* new_xyz_batch_cnt, const int32_t* idx, const uint8_t* empty_mask,
                                        const float* rel, const float* grad_z, const float* y, const float* mean,
                                        const float* invstd, const float* gamma, const float* beta, const float* dbeta,
                                        const float* dgamma, float* grad_P, float* part, void* stream);
int crb_group_affine_rows_grad_stack(int B, int64_t M, int H, int nsample, const int32_t* xyz_batch_cnt,
                                     const int32_t* new_xyz_batch_cnt, const int32_t* idx, const uint8_t* empty_mask,
                                     const float* rel, const float* grad_out, float* grad_P, float* part, void* stream);
/* inference-only fused set abstraction: replaces the body of StackSAModuleMSG.forward
 * (pointnet2_modules.py:73-112: QueryAndGroup -> 2 x [Conv2d 1x1 + BatchNorm2d + ReLU] -> max_pool2d over nsample) for one
 * radius. BN is folded by the caller; layer 1 is split as W1 [dxyz ; f] = W1x dxyz + P[row], P = features @ W1f^T (N,h1).
 * W1x (3,h1), b1 (h1), W2 (h1,h2) [k][n], b2 (h2); idx / empty_mask as for crb_query_group_stack; writes
 * out[m*out_stride + n], n < h2 (out may point into a wider (M, sum h2) buffer). h1,h2 in {16,32,64}. */
int crb_sa_mlp2_max_supported(int h1, int h2);
int crb_sa_mlp2_max_stack(int B, int64_t M, int nsample, int h1, int h2, const float* xyz,
                          const int32_t* xyz_batch_cnt, const float* P, const float* new_xyz,
                          const int32_t* new_xyz_batch_cnt, const int32_t* idx, const uint8_t* empty_mask,
                          const float* W1x, const float* b1, const float* W2, const float* b2, float* out,
                          int out_stride, void* stream);
/* TRAINING-mode fused set abstraction of one radius (round 5): the body of StackSAModuleMSG.forward for a two-layer shared MLP
 * (pointnet2_modules.py:90-108: Conv2d 1x1 -> BatchNorm2d -> ReLU -> Conv2d 1x1 -> BatchNorm2d -> ReLU -> max_pool2d over nsample;
 * RoI-grid pooling: pvrcnn_head.py:102-113) without any (M*nsample, h) activation in memory. Train-mode BatchNorm needs the
 * full-tensor statistics of a layer before the next one can run, so the forward is three recompute passes over the ball-query
 * result:  (0) crb_group_affine_rows_stats_stack with out = rel = NULL -> slab sums of y1 -> crb_bn_relu_forward_partials(z = NULL);
 * (A) crb_sa_mlp2_train_stats -> wave_sums (crb_sa_mlp2_train_waves(M), 2, h2): column sums of y2, y2^2 per wave ->
 * crb_bn_relu_forward_partials(z = NULL);  (B) crb_sa_mlp2_train_max -> out[m*out_row_stride + c] = max over the samples of
 * relu(bn2(y2)), arg (M,h2) = the sample that attains it (lowest index on ties, as crb_bn_relu_max_forward), y_sel (M,h2) = y2 there.
 * W1x (3,h1), W2 (h2,h1) = the second conv's weight, P = features @ W1f^T (N,h1), mean / invstd = batch statistics,
 * gamma / beta = the BatchNorm parameters. h1, h2 in {16,32,64}, nsample a multiple of 16, M*nsample < 2^31. */
int crb_sa_mlp2_train_supported(int h1, int h2, int nsample);
int64_t crb_sa_mlp2_train_waves(int64_t M);
int crb_sa_mlp2_train_stats(int B, int64_t M, int nsample, int h1, int h2, const float* xyz,
                            const int32_t* xyz_batch_cnt, const float* P, const float* new_xyz,
                            const int32_t* new_xyz_batch_cnt, const int32_t* idx, const uint8_t* empty_mask,
                            const float* W1x, const float* mean1, const float* invstd1, const float* gamma1,
                            const float* beta1, const float* W2, float* wave_sums, void* stream);
int crb_sa_mlp2_train_max(int B, int64_t M, int nsample, int h1, int h2, const float* xyz,
                          const int32_t* xyz_batch_cnt, const float* P, const float* new_xyz,
                          const int32_t* new_xyz_batch_cnt, const int32_t* idx, const uint8_t* empty_mask,
                          const float* W1x, const float* mean1, const float* invstd1, const float* gamma1,
                          const float* beta1, const float* W2, const float* mean2, const float* invstd2,
                          const float* gamma2, const float* beta2, float* out, int64_t out_row_stride, int32_t* arg,
                          float* y_sel, void* stream);
/* backward of the above (autograd of the same module lines). Caller first reduces dbeta2 / dgamma2 from the selected entries
 * (crb_bn_relu_max_backward_sums on y_sel). This pass recomputes y1, z1, y2, forms dy2 (BatchNorm-2 backward of the max's scatter)
 * in registers, and produces: grad_z1_masked (M*nsample, h1) = (dy2 W2) [z1 > 0], dsums1 (2, h1) = its column sums and its column
 * sums times xhat1 (dbeta1, dgamma1), dW2 (h2, h1). The first layer's own backward then is
 * crb_group_affine_rows_grad_bn_recompute_stack. workspace: crb_sa_mlp2_train_backward_workspace_floats(M, h1, h2) floats. */
int64_t crb_sa_mlp2_train_backward_workspace_floats(int64_t M, int h1, int h2);
int crb_sa_mlp2_train_backward(int B, int64_t M, int nsample, int h1, int h2, const float* xyz,
                               const int32_t* xyz_batch_cnt, const float* P, const float* new_xyz,
                               const int32_t* new_xyz_batch_cnt, const int32_t* idx, const uint8_t* empty_mask,
                               const float* W1x, const float* mean1, const float* invstd1, const float* gamma1,
                               const float* beta1, const float* W2, const float* mean2, const float* invstd2,
                               const float* gamma2, const float* beta2, const float* grad_out,
                               int64_t grad_row_stride, const int32_t* arg, const float* dbeta2, const float* dgamma2,
                               float* grad_z1_masked, float* dsums1, float* dW2, float* workspace,
                               int64_t workspace_floats, void* stream);
/* crb_group_affine_rows_grad_bn_stack without the saved forward tensors: y (the first conv's output) and rel are recomputed from
 * xyz / new_xyz / P / W1x exactly as crb_group_affine_rows_stack forms them. */
int crb_group_affine_rows_grad_bn_recompute_stack(int B, int64_t M, int H, int nsample, const float* xyz,
                                                  const int32_t* xyz_batch_cnt, const float* P, const float* new_xyz,
                                                  const int32_t* new_xyz_batch_cnt, const int32_t* idx,
                                                  const uint8_t* empty_mask, const float* W1x, const float* grad_z,
                                                  const float* mean, const float* invstd, const float* gamma,
                                                  const float* beta, const float* dbeta, const float* dgamma,
                                                  const int32_t* sorted_pair, const int32_t* sorted_row, int64_t n_src,
                                                  float* grad_P, float* part, void* stream);
/* deterministic form of the call above (torch.use_deterministic_algorithms in the host mirror): what leaves a slab is added into
 * grad_P_fixed (n_src, H) int64, pre-zeroed, as round(value * scale) with 64-bit integer atomics - the sums do not depend on the
 * order in which the slabs arrive; the caller converts back (grad_P = grad_P_fixed / scale). scale > 0: 2^40 / (a power of two >=
 * max |grad_z| * max |gamma * invstd|) keeps 2^-40 of that magnitude per addend and overflows only past 2^22 times it. */
int crb_group_affine_rows_grad_bn_recompute_stack_fixed(int B, int64_t M, int H, int nsample, const float* xyz,
                                                        const int32_t* xyz_batch_cnt, const float* P, const float* new_xyz,
                                                        const int32_t* new_xyz_batch_cnt, const int32_t* idx,
                                                        const uint8_t* empty_mask, const float* W1x, const float* grad_z,
                                                        const float* mean, const float* invstd, const float* gamma,
                                                        const float* beta, const float* dbeta, const float* dgamma,
                                                        const int32_t* sorted_pair, const int32_t* sorted_row, int64_t n_src,
                                                        int64_t* grad_P_fixed, float scale, float* part, void* stream);
/* optional operand of the calls above (sorted_pair / sorted_row, NULL = pair order): the (query, sample) pairs in SOURCE-ROW order,
 * stable (pairs of one row keep their order), pairs of empty balls last with sorted_row = n_src (the number of source rows). With it
 * the kernel adds the pairs of a row inside LDS and issues one row of atomics per run of a 64-pair slab instead of one per distinct
 * row of a 16-pair segment: for layers where many balls share a row (the RoI-grid scales of PV-RCNN: 7 M pairs onto 32 k keypoints,
 * 1.89 -> 0.81 ms + 0.19 ms for the sort; not for the voxel levels, where the sort costs more than it gains). Same sums, another
 * order of the float additions into grad_P. workspace: crb_pair_sort_workspace_bytes(M, nsample). */
int64_t crb_pair_sort_workspace_bytes(int64_t M, int nsample);
int crb_pair_sort_by_source(int B, int64_t M, int nsample, const int32_t* xyz_batch_cnt, const int32_t* new_xyz_batch_cnt,
                            const int32_t* idx, const uint8_t* empty_mask, int64_t n_src, int32_t* sorted_pair,
                            int32_t* sorted_row, void* workspace, int64_t workspace_bytes, void* stream);
/* rows per frame of a stacked tensor from its frame-index column (replaces the per-frame `(xyz_bs_idxs == bs_idx).sum()` of
 * voxel_set_abstraction.py:321-323, :365-367 and pvrcnn_head.py:96-98): key = the column (f32 if key_is_float else i32), `stride` elements
 * from row to row, n rows, NON-DECREASING (the stacked layout: rows of a frame together, frames in order) -> counts (B) i32.
 * Values outside [0, B) are not counted. One launch, no atomics. */
int crb_sorted_key_counts(const void* key, int key_is_float, int64_t stride, int64_t n, int B, int32_t* counts, void* stream);
/* voxel centres (common_utils.get_voxel_centers, pcdet/utils/common_utils.py:63-80): coords_zyx (n rows of 3 i32 [z,y,x], row_stride
 * elements apart: columns 1..3 of the (n,4) index tensor) -> out (n,3) xyz = (coord + 0.5) * scaled_voxel_size + range_min;
 * scaled_voxel_size[3] = voxel size x downsample factor (f32 product), range_min[3]: HOST arrays. */
int crb_voxel_centers(const int32_t* coords_zyx, int64_t row_stride, int64_t n, const float* scaled_voxel_size, const float* range_min,
                      float* out, void* stream);
/* xyz (B,n,3) -> out_idx (B,m); first pick is index 0; ties resolved like the reference kernel (see source).
 * temp: (B,n) f32 scratch for the running distances (the reference's `temp` argument); only needed for n > 40960
 * (below that the distances stay in registers) and may be NULL otherwise. */
int crb_farthest_point_sample(int B, int n, int m, const float* xyz, float* temp, int32_t* out_idx, void* stream);
int crb_three_nn_stack(int B, int64_t N, const float* unknown, const int32_t* unknown_batch_cnt,
                       const float* known, const int32_t* known_batch_cnt, float* dist2, int32_t* idx,
                       void* stream);
int crb_three_interpolate_stack(int64_t N, int C, const float* features, const int32_t* idx,
                                const float* weight, float* out, void* stream);
int crb_three_interpolate_grad_stack(int64_t N, int C, const float* grad_out, const int32_t* idx,
                                     const float* weight, float* grad_features, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a15  points-in-boxes and RoI-aware pooling
 * replaces: pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:172-177 (points_in_boxes_gpu :94-119, forward :46-70,
 *           backward :72-92).  boxes (B,T,7), pts (B,M,3) -> box_idx_of_points (B,M) i32, -1 = background.
 * pool: rois (N,7), pts (P,3), pts_feature (P,C); pts_idx_of_voxels (N,ox,oy,oz,max_pts) i32 and pooled_features
 * (N,ox,oy,oz,C) must be zero-filled by the caller; argmax (N,ox,oy,oz,C) i32. pool_method 0 = max, 1 = avg.
 * ---------------------------------------------------------------------------------------------- */
int crb_points_in_boxes(int B, int T, int M, const float* boxes, const float* pts, int32_t* box_idx_of_points,
                        void* stream);
/* GT point statistics of the CRB-patched post-processing, all frames and classes in two launches.
 * replaces: the per-frame / per-class loop of pcdet/models/detectors/detector3d_template.py:236-268 (points_in_boxes_gpu
 *           per class + `(idx == i).sum()` per unique index + torch.mean / median / var on the host side).
 * pts (N,stride) f32 rows [b,x,y,z,...] frame-sorted, frame_offsets (B+1) i32, gt_boxes (B,G,8) rows
 * [x,y,z,dx,dy,dz,heading,label] (label 0 = padding) -> stats (B,C,5) f32 {num_bbox, n_counted, mean, median, variance}
 * of the per-box point counts (boxes owning no point are not counted; when a frame has no background point for a class
 * the first counted box is dropped, as the reference's `[1:]` does). workspace: the first B*G i32 hold the per-box
 * counts afterwards (then B*C background counts). */
int64_t crb_gt_point_stats_workspace_bytes(int B, int G, int C);
int crb_gt_point_stats(int B, int G, int C, int64_t N, int stride, const float* pts, const int32_t* frame_offsets,
                       const float* gt_boxes, float* stats, void* workspace, int64_t workspace_bytes, void* stream);
int crb_roiaware_pool3d_forward(int N, int P, int C, int max_pts_each_voxel, int out_x, int out_y, int out_z,
                                const float* rois, const float* pts, const float* pts_feature,
                                int32_t* argmax, int32_t* pts_idx_of_voxels, float* pooled_features,
                                int pool_method, void* stream);
int crb_roiaware_pool3d_backward(int N, int C, int max_pts_each_voxel, int out_x, int out_y, int out_z,
                                 const int32_t* pts_idx_of_voxels, const int32_t* argmax,
                                 const float* grad_out, float* grad_in, int pool_method, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a28  CRB stage 3: greedy point-cloud-density balancing
 * replaces: the sklearn KernelDensity / scipy.stats.entropy triple loop of
 *           pcdet/query_strategies/crb_sampling.py:276-331 (SELECT_NUMS x candidates x classes KDE fits on the CPU).
 * densities (n_candidates,dmax) f32 / labels (n_candidates,dmax) i32 (1..num_class, 0 = padding): predicted-box point
 * densities of the K2*N candidate frames in candidate order; xaxis/prior (num_class,400) f64 device: evaluation axis
 * and (un-normalised) uniform prior per class; order (select_nums) i32: picked candidate indices, first pick = 0,
 * -1 padded; best_scores (select_nums) f64 or NULL. f64 arithmetic. No synchronisation.
 * ---------------------------------------------------------------------------------------------- */
int64_t crb_density_greedy_workspace_bytes(int n_candidates, int num_class);
int crb_density_greedy(const float* densities, const int32_t* labels, int n_candidates, int dmax,
                       int num_class, const double* xaxis, const double* prior, double bandwidth,
                       int select_nums, int32_t* order, double* best_scores, void* workspace,
                       int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a5  fused BatchNorm1d (+ReLU) over sparse-tensor rows
 * replaces: nn.BatchNorm1d(eps=1e-3, momentum=0.01) + nn.ReLU after every sparse conv
 *           (pcdet/models/backbones_3d/spconv_backbone.py:21-25,73). x,z,dx,dz (n,C) f32, C % 4 == 0 and
 *           (C/4) divides 256 (C = 16,32,64,128,...). Training forward returns batch mean / biased var / invstd;
 *           running_mean / running_var (nullable) are updated in the same launch: r = (1-momentum) r + momentum batch,
 *           variance unbiased (n/(n-1)), as nn.BatchNorm does; num_batches_tracked (nullable, one int64) += 1 likewise.
 *           tickets (nullable): crb_bn_ticket_ints() int32 in device memory, ZERO before the first call and left zero by
 *           every call; with it the statistics launch also reduces its per-block partials ("last block done", same
 *           summation order: bit-identical results) and a call is two launches instead of three. One ticket area serves
 *           any number of calls on ONE stream (stream order keeps them apart); calls that may run concurrently on
 *           different streams need their own. NULL = the partials are reduced by a launch of their own.
 * ---------------------------------------------------------------------------------------------- */
int64_t crb_bn_workspace_bytes(int64_t n, int C);
int crb_bn_ticket_ints(void);
int crb_bn_relu_forward(const float* x, int64_t n, int C, const float* gamma, const float* beta, float eps,
                        int relu, float* z, int64_t z_row_stride, float* mean, float* var, float* invstd,
                        float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                        void* workspace, int64_t workspace_bytes, int32_t* tickets, void* stream);
/* training forward whose statistics come from slab sums the producer of x wrote (slab_sums (n_slabs, 2, C): column sums of x
 * and x^2 over consecutive row slabs covering all n rows): same outputs as crb_bn_relu_forward up to the rounding of another
 * summation order, without the statistics pass over x. */
int crb_bn_relu_forward_partials(const float* x, int64_t n, int C, const float* slab_sums, int64_t n_slabs, const float* gamma,
                                 const float* beta, float eps, int relu, float* z, int64_t z_row_stride, float* mean, float* var,
                                 float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                 float momentum, void* workspace, int64_t workspace_bytes, int32_t* tickets, void* stream);
/* z == NULL in crb_bn_relu_forward / crb_bn_relu_forward_partials: statistics only (mean / var / invstd, running statistics and the
 * batch counter as usual), no apply pass — for a consumer that applies the normalisation itself:
 * crb_bn_affine_table writes (scale, shift) = (gamma * invstd, beta - mean * gamma * invstd) per channel, interleaved (C, 2), the
 * table crb_conv3x3_winograd2_bnrelu_nhwc / crb_winograd2_wgrad_bnrelu take. */
int crb_bn_affine_table(const float* mean, const float* invstd, const float* gamma, const float* beta, int C, float* out_c2,
                        void* stream);
int crb_bn_relu_apply(const float* x, int64_t n, int C, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, int relu, float* z, int64_t z_row_stride, void* stream);
/* z_row_stride / dz_row_stride (floats, 0 = C): z may be a channel slice of a wider row-major buffer (the BEV backbone
 * writes its two up-sampled branches straight into the concatenated (N*H*W, 512) map, and their backward reads the matching
 * slices of its gradient: no torch.cat copy, no .contiguous() copies of the gradient slices). */
/* dx may be NULL: only dgamma / dbeta are computed (a consumer that applies the BatchNorm backward while it reads dz) */
int crb_bn_relu_backward(const float* x, const float* dz, int64_t dz_row_stride, int64_t n, int C, const float* mean,
                         const float* invstd, const float* gamma, const float* beta, int relu, float* dx,
                         float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes, int32_t* tickets,
                         void* stream);
/* training BatchNorm + ReLU over groups*ns rows followed by the max over every group of ns consecutive rows: the tail of
 * a StackSAModuleMSG scale (pointnet2_modules.py:96-103: BatchNorm2d, ReLU, F.max_pool2d over nsample) on the row-major
 * (M*nsample, C) layout. The normalised matrix is not materialised: zmax (groups, out_row_stride; 0 = C) and
 * arg (groups, C) = first row of the group attaining the max are the outputs; the backward takes the gradient of zmax
 * (groups, gz_row_stride) and writes the dense dx. workspace: crb_bn_workspace_bytes(groups*ns, C). */
int crb_bn_relu_max_forward(const float* x, int64_t groups, int ns, int C, const float* gamma, const float* beta, float eps,
                            float* zmax, int64_t out_row_stride, int32_t* arg, float* mean, float* var, float* invstd,
                            float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                            void* workspace, int64_t workspace_bytes, int32_t* tickets, void* stream);
int crb_bn_relu_max_backward(const float* x, const float* gz, int64_t gz_row_stride, const int32_t* arg, int64_t groups,
                             int ns, int C, const float* mean, const float* invstd, const float* gamma, const float* beta,
                             float* dx, float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes,
                             int32_t* tickets, void* stream);
/* the reduction half of crb_bn_relu_max_backward alone, for a caller that kept only the selected entries: x_sel (groups, C) = the
 * BatchNorm input at each group's arg row (crb_sa_mlp2_train_max's y_sel), gz as above -> dbeta = sum gz [z > 0],
 * dgamma = sum gz [z > 0] xhat. workspace: crb_bn_workspace_bytes(groups, C). */
int crb_bn_relu_max_backward_sums(const float* x_sel, const float* gz, int64_t gz_row_stride, int64_t groups, int C,
                                  const float* mean, const float* invstd, const float* gamma, const float* beta, float* dgamma,
                                  float* dbeta, void* workspace, int64_t workspace_bytes, int32_t* tickets, void* stream);
/* Per-frame statistics (batched CRB stage 2: G frames per train-mode pass, every BatchNorm layer normalising each frame
 * with that frame's own batch statistics like G bs=1 passes, crb_sampling.py:174-212): four launches for all frames
 * (partial sums / finalize / running update / apply, a frame's rows cut into the blocks a single-frame call would use and
 * summed in the same order: bit-identical to n_frames calls of the forward above on the frames' row ranges). frame_row_offsets is a HOST int64[n_frames+1] array; the batch statistics are scratch, running
 * statistics are updated once per frame in frame order. Forward only (stage 2 differentiates the RoI-head FC stack only). */
int64_t crb_bn_frames_workspace_bytes(int n_frames, int64_t max_rows_per_frame, int C);
int crb_bn_relu_forward_frames(const float* x, int n_frames, const int64_t* frame_row_offsets, int C, const float* gamma,
                               const float* beta, float eps, int relu, float* z, int64_t z_row_stride,
                               float* running_mean, float* running_var, float momentum, void* workspace,
                               int64_t workspace_bytes, void* stream);
int crb_bn_relu_max_forward_frames(const float* x, int n_frames, int64_t groups_per_frame, int ns, int C,
                                   const float* gamma, const float* beta, float eps, float* zmax, int64_t out_row_stride,
                                   int32_t* arg, float* running_mean, float* running_var, float momentum, void* workspace,
                                   int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a10  anchor target assignment (nearest-BEV IoU + thresholds + residual box encoding)
 * replaces: AxisAlignedTargetAssigner.assign_targets (pcdet/models/dense_heads/target_assigner/
 *           axis_aligned_target_assigner.py:36-210), box_utils.boxes3d_nearest_bev_iou (pcdet/utils/box_utils.py:272-298),
 *           ResidualCoder.encode_torch (pcdet/utils/box_coder_utils.py:13-43); POS_FRACTION < 0 branch, single head.
 * anchors (A,7) in the head's flattened order with anchor_cls (A) i32 (1-based class); gt_boxes (B,G,8) zero padded,
 * gt_valid (B,G) u8; matched_thr / unmatched_thr: device f32 arrays indexed by class id (entry 0 unused).
 * out: labels (B,A) i32 {-1,0,class}, reg_targets (B,A,7), reg_weights (B,A).
 * ---------------------------------------------------------------------------------------------- */
int64_t crb_assign_targets_workspace_bytes(int B, int A, int G);
int crb_assign_targets(const float* anchors, const int32_t* anchor_cls, int A, const float* gt_boxes,
                       const uint8_t* gt_valid, int B, int G, const float* matched_thr,
                       const float* unmatched_thr, int32_t* labels, float* reg_targets, float* reg_weights,
                       void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a11  RPN losses of the anchor head: sigmoid focal classification + weighted smooth-L1 regression (with the sine
 *      difference of the heading) + direction-bin cross entropy, forward and gradient
 * replaces: AnchorHeadTemplate.get_cls_layer_loss / add_sin_difference / get_direction_target / get_box_reg_layer_loss
 *           (pcdet/models/dense_heads/anchor_head_template.py:101-214), SigmoidFocalClassificationLoss,
 *           WeightedSmoothL1Loss, WeightedCrossEntropyLoss (pcdet/utils/loss_utils.py:9-188)
 * cls_preds (B,A,num_class) logits, box_preds (B,A,7), dir_preds (B,A,num_dir_bins) or NULL (no direction classifier),
 * labels (B,A) i32 {-1 ignored, 0 background, c > 0 class; num_class == 1: any c > 0 is the class}, reg_targets (B,A,7) as
 * written by crb_assign_targets (NaN entries = "no target": zero loss and gradient), anchors (A,7) (heading column only).
 * forward : loss (B,3) = per frame {cls, loc, dir} losses, each = weight * sum over anchors / max(npos,1); npos (B) f32 =
 *           positives of the frame. The reference's scalar losses are loss.sum(0) / B; its reduce=False variants are the rows.
 * backward: gradient of sum_b sum_k grad_loss[b,k] * loss[b,k] w.r.t. the three prediction tensors (every element written).
 * Sums are taken in a fixed order: bit-reproducible. Limits: num_class <= 8, num_dir_bins <= 8 (else CRB_ERR_ARG).
 * ---------------------------------------------------------------------------------------------- */
typedef struct CrbRpnLossCfg {
  float alpha, gamma;            /* focal loss (0.25, 2.0) */
  float beta;                    /* smooth-L1 knee (1/9) */
  float dir_offset;              /* MODEL.DENSE_HEAD.DIR_OFFSET */
  float code_weights[7];         /* LOSS_CONFIG.LOSS_WEIGHTS.code_weights */
  float cls_weight, loc_weight, dir_weight;
  int32_t num_class, num_dir_bins;
} CrbRpnLossCfg;
int64_t crb_rpn_loss_workspace_bytes(int B, int A);
int crb_rpn_loss_forward(const float* cls_preds, const float* box_preds, const float* dir_preds, const int32_t* labels,
                         const float* reg_targets, const float* anchors, int B, int A, const CrbRpnLossCfg* cfg,
                         float* loss, float* npos, void* workspace, int64_t workspace_bytes, void* stream);
int crb_rpn_loss_backward(const float* cls_preds, const float* box_preds, const float* dir_preds, const int32_t* labels,
                          const float* reg_targets, const float* anchors, int B, int A, const CrbRpnLossCfg* cfg,
                          const float* npos, const float* grad_loss, float* d_cls, float* d_box, float* d_dir, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a21  proposal layer around its NMS (csrc/proposal_layer.hip)
 * crb_decode_selected_anchors replaces AnchorHeadTemplate.generate_predicted_boxes for the anchors the proposal layer keeps
 *   (pcdet/models/dense_heads/anchor_head_template.py:238-285 with ResidualCoder.decode_torch, pcdet/utils/box_coder_utils.py:45-73,
 *   and the direction-bin correction through common_utils.limit_period): box_preds (B,A,7), dir_preds (B,A,num_dir_bins) or NULL,
 *   anchors (A,7), anchor_idx (B,k) i64 -> out (B,k,7) = the rows the full decode holds at those indices, bit for bit.
 * crb_proposal_finish replaces the gathers of RoIHeadTemplate.proposal_layer behind the class-agnostic NMS
 *   (pcdet/models/roi_heads/roi_head_template.py:73-108): keep (B,post) i32 (-1 = padding, positions in the top-k order), top_idx (B,k)
 *   i64 anchor indices, top_boxes (B,k,box_row_stride), scores (B,A), labels (B,A) i64 (0-based argmax), cls_preds (B,A,num_class)
 *   -> rois (B,post,box_row_stride) zero padded, roi_scores (B,post), roi_labels (B,post) i64 (1-based; padding = 1 as in the
 *   reference), full_cls_scores (B,post,num_class).
 * ---------------------------------------------------------------------------------------------- */
int crb_decode_selected_anchors(const float* box_preds, const float* dir_preds, const float* anchors, const int64_t* anchor_idx, int B,
                                int64_t A, int k, int num_dir_bins, float dir_offset, float dir_limit_offset, float* out, void* stream);
int crb_proposal_finish(const int32_t* keep, const int64_t* top_idx, const float* top_boxes, const float* scores, const int64_t* labels,
                        const float* cls_preds, int B, int64_t A, int k, int post, int box_row_stride, int num_class, float* rois,
                        float* roi_scores, int64_t* roi_labels, float* full_cls_scores, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a20  point head in training (csrc/point_head.hip)
 * crb_point_labels replaces the label arithmetic of PointHeadTemplate.assign_stack_targets behind its two points_in_boxes_gpu calls
 *   (pcdet/models/dense_heads/point_head_template.py:49-129, set_ignore_flag mode of PointHeadSimple): inner / outer (B*M) i32 = first
 *   containing ground truth / enlarged ground truth of every point or -1, gt_boxes (B,G,gt_row_stride >= 8, class last)
 *   -> labels (B*M) i64: the box's class (1 when num_class == 1) inside a box, -1 in the enlarged shell only, 0 elsewhere.
 * crb_point_focal_loss replaces PointHeadTemplate.get_cls_layer_loss with SigmoidFocalClassificationLoss (:131-155,
 *   pcdet/utils/loss_utils.py:9-72) and its autograd: preds (n, num_class) logits, labels (n) i64 -> loss[3] = {point_loss_cls
 *   (x loss_weight, normalised by max(positives, 1)), positives, point_loss_cls again (the scalar a caller differentiates)},
 *   d_preds (n, num_class) = d loss / d preds. One workgroup, sums in a fixed order.
 * ---------------------------------------------------------------------------------------------- */
int crb_point_labels(const int32_t* inner, const int32_t* outer, const float* gt_boxes, int B, int64_t M, int G, int gt_row_stride,
                     int num_class, int64_t* labels, void* stream);
int crb_point_focal_loss(const float* preds, const int64_t* labels, int64_t n, int num_class, float alpha, float gamma,
                         float loss_weight, float* loss, float* d_preds, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a22 / a24  second-stage losses and the canonical transformation of the sampled ground truths (csrc/rcnn_loss.hip)
 * replaces: RoIHeadTemplate.get_box_cls_layer_loss (BinaryCrossEntropy; pcdet/models/roi_heads/roi_head_template.py:261-285),
 *           get_box_reg_layer_loss (smooth-l1 + CORNER_LOSS_REGULARIZATION, the branch without reg_sample_targets; :142-259) with
 *           ResidualCoder.encode_torch / decode_torch (pcdet/utils/box_coder_utils.py:13-73), get_corner_loss_lidar
 *           (pcdet/utils/loss_utils.py:209-232), boxes_to_corners_3d (pcdet/utils/box_utils.py:28-55); and the part of
 *           RoIHeadTemplate.assign_targets after the sampling (:118-138). ~270 torch launches of a PV-RCNN step as two.
 * crb_rcnn_loss: n sampled RoIs; rcnn_cls (n) logits, rcnn_reg (n,7), cls_labels (n) f32 or i64 (< 0 = ignored; soft labels in
 * [0,1] for CLS_SCORE_TYPE roi_iou), reg_valid_mask (n) i64, rois (n,7), gt_of_rois (n, gt_row_stride) in the RoI frame,
 * gt_of_rois_src the same boxes in LiDAR coordinates (corner loss; NULL when cfg->corner == 0)
 * -> loss[7] = {rcnn_loss_cls, rcnn_loss_reg, rcnn_loss_corner, rcnn_loss (their sum), foreground RoIs, valid RoIs, rcnn_loss again
 *    (a second home for the scalar a caller differentiates)}, each loss already multiplied by its LOSS_WEIGHTS entry and divided by
 *    max(count, 1);
 *    d_cls (n), d_reg (n,7) = d rcnn_loss / d (rcnn_cls, rcnn_reg); reg_targets (n,7) (forward_ret_dict['rcnn_reg_gt']) or NULL.
 * One workgroup, sums in a fixed order: bit-reproducible. code size 7 only.
 * crb_roi_canonical_targets: rois (n, roi_row_stride >= 7), gt_of_rois (n, gt_row_stride >= 7) -> out (n, gt_row_stride).
 * ---------------------------------------------------------------------------------------------- */
typedef struct CrbRcnnLossCfg {
  float beta;                    /* smooth-L1 knee of the regression loss (1/9); the corner loss uses 1 */
  float code_weights[7];         /* LOSS_CONFIG.LOSS_WEIGHTS.code_weights */
  float cls_weight, reg_weight, corner_weight;   /* rcnn_cls_weight, rcnn_reg_weight, rcnn_corner_weight */
  int32_t corner;                /* CORNER_LOSS_REGULARIZATION */
} CrbRcnnLossCfg;
int crb_rcnn_loss(const float* rcnn_cls, const float* rcnn_reg, const void* cls_labels, int labels_are_int64,
                  const int64_t* reg_valid_mask, const float* rois, const float* gt_of_rois, const float* gt_of_rois_src,
                  int gt_row_stride, int64_t n, const CrbRcnnLossCfg* cfg, float* loss, float* d_cls, float* d_reg,
                  float* reg_targets, void* stream);
int crb_roi_canonical_targets(const float* rois, int roi_row_stride, const float* gt_of_rois, int gt_row_stride, int64_t n,
                              float* out, void* stream);
/* grid points of the RoI-grid pooling (PVRCNNHead.get_global_grid_points_of_roi + get_dense_grid_points, pvrcnn_head.py:116-141):
 * rois (n, roi_row_stride >= 7) -> out (n, grid_size^3, 3): ((i + 0.5) / G) * size - size / 2 per axis (x slowest, z fastest), turned by
 * the heading about z, moved to the centre. */
int crb_roi_grid_points(const float* rois, int roi_row_stride, int64_t n, int grid_size, float* out, void* stream);
/* second-stage box decode (RoIHeadTemplate.generate_predicted_boxes, roi_head_template.py:335-359): rois (n, roi_row_stride >= 7),
 * box_preds (n,7) residuals -> out (n,7) boxes in LiDAR coordinates (ResidualCoder.decode_torch against the RoI as anchor with centre
 * 0, the centre turned by the RoI's heading and moved to the RoI's centre). */
int crb_rcnn_decode_boxes(const float* rois, int roi_row_stride, const float* box_preds, int64_t n, float* out, void* stream);
/* CRB stage-1 records behind the final NMS (Detector3DTemplate.post_processing's per-frame selection, detector3d_template.py:186-234,
 * and the label entropy of crb_sampling.py:86-94), one workgroup per frame: sel (B,P) i64 indices into the N boxes of the frame, valid
 * (B,P) u8, box_preds (B,N,box_row_stride), cls_confs (B,N), label_preds (B,N) i64 (1-based), full_cls_scores (B,N,full_classes) or
 * NULL -> pred_boxes (B,P,box_row_stride), pred_scores (B,P), pred_labels (B,P) i64, pred_logits (B,P,full_classes), all zero where
 * not valid; entropy (B) = Shannon entropy of the label histogram of the valid boxes, absent classes counted as 1, 0 without boxes.
 * crb_box_point_density: first_box (B,M) i32 = first containing predicted box of every point (crb_points_in_boxes) -> density (B,P) =
 * points / max(volume, 1e-12) of the valid boxes, 0 elsewhere (P <= 2048). */
int crb_record_rows(const int64_t* sel, const uint8_t* valid, const float* box_preds, int box_row_stride, const float* cls_confs,
                    const int64_t* label_preds, const float* full_cls_scores, int full_classes, int B, int N, int P, int num_class,
                    float* pred_boxes, float* pred_scores, int64_t* pred_labels, float* pred_logits, float* entropy, void* stream);
int crb_box_point_density(const int32_t* first_box, const float* pred_boxes, int box_row_stride, const uint8_t* valid, int B, int M, int P,
                          float* density, void* stream);
/* RoI sampling for the second stage: one workgroup per frame (csrc/rcnn_loss.hip)
 * replaces: ProposalTargetLayer.forward / sample_rois_for_rcnn / subsample_rois / get_max_iou_with_same_class
 *           (pcdet/models/roi_heads/target_assigner/proposal_target_layer.py:15-228): per frame the (same-class) maximum IoU of every
 *           proposal, foreground / hard / easy background sets, FG_RATIO and HARD_BG_RATIO quotas, foreground without replacement in
 *           random order, background with replacement, the gathers, reg_valid_mask and rcnn_cls_labels. The random numbers are
 *           INPUTS: u_perm (B,R) orders the foreground, u_slot (B,P) draws with replacement (floor(u * n)) - the reference's
 *           np.random / torch.randint stream is not reproduced.
 * rois (B,R,roi_row_stride >= 7), roi_scores (B,R), roi_labels (B,R) i64, gt_boxes (B,G,gt_row_stride >= 8, class in the last column,
 * zero rows = padding), iou (B*R, B*G) = crb_boxes_iou3d of all proposals against all ground truths (frame b uses its diagonal block)
 * -> sampled (B,P) i64 proposal indices, out_rois (B,P,roi_row_stride), out_gt (B,P,gt_row_stride), out_iou / out_scores (B,P),
 *    out_labels (B,P) i64, reg_valid_mask (B,P) i64, cls_labels (B,P) f32 (score_type 0 = roi_iou) or i64 (1 = cls, -1 ignored).
 * R <= 1024, else CRB_ERR_UNSUPPORTED. */
typedef struct CrbRoiSamplerCfg {
  int32_t roi_per_image;         /* ROI_PER_IMAGE = P */
  int32_t fg_quota;              /* round(FG_RATIO * P) */
  int32_t by_class;              /* SAMPLE_ROI_BY_EACH_CLASS */
  int32_t score_type;            /* CLS_SCORE_TYPE: 0 roi_iou, 1 cls */
  float fg_thresh;               /* min(REG_FG_THRESH, CLS_FG_THRESH) */
  float reg_fg_thresh, cls_fg_thresh, cls_bg_thresh, cls_bg_thresh_lo, hard_bg_ratio;
  float soft_den;                /* CLS_FG_THRESH - CLS_BG_THRESH */
} CrbRoiSamplerCfg;
int crb_roi_sample_targets(const float* rois, int roi_row_stride, const float* roi_scores, const int64_t* roi_labels,
                           const float* gt_boxes, int gt_row_stride, const float* iou, const float* u_perm, const float* u_slot,
                           int B, int R, int G, const CrbRoiSamplerCfg* cfg, int64_t* sampled, float* out_rois, float* out_gt,
                           float* out_iou, float* out_scores, int64_t* out_labels, int64_t* reg_valid_mask, void* cls_labels,
                           void* stream);

/* ------------------------------------------------------------------------------------------------
 * a7  3x3 stride-1 pad-1 convolution on channels_last maps as Winograd F(2x2,3x3) on the f32 MFMA (csrc/winograd_conv2.hip)
 * replaces: torch.nn.Conv2d(C, C, 3, padding=1) of the BEV backbone (pcdet/models/backbones_2d/base_bev_backbone.py:24-41;
 *           MIOpen's f32 implicit GEMM on this stack). 2.25x fewer multiplications than the direct convolution, results equal to
 *           it up to f32 rounding of the transforms (4e-7 of the output scale against an f64 convolution).
 * x (N,H,W,Cin) f32 NHWC, y (N,H,W,Cout); bias (Cout) or NULL, relu 0/1: epilogue on the output. The input gradient of the same
 * layer is the same call on dy with the flipped, transposed weights. Two waves per SIMD (128 accumulators each), the raw input
 * block and the weight block brought into LDS by LDS-DMA once per workgroup and chunk, the input transform read from LDS;
 * workgroup = 16 x 4 tiles of one spatial block x 64 output channels. Weight image: crb_winograd2_weights writes U in the order
 * the kernel's LDS-DMA copies it ([Cout/64][Cin/8][LDS image of a chunk]), crb_winograd2_weights_bytes floats*4.
 * crb_winograd2_supported: Cin % 8 == 0, Cout % 64 == 0, H >= 5. (The first design of round 3, crb_conv3x3_winograd_nhwc, lives in
 * the measurement library: include/crb_hip_measure.h.) */
int crb_winograd2_supported(int cin, int cout, int H, int W);
int64_t crb_winograd2_weights_bytes(int cin, int cout);
int crb_winograd2_weights(const float* g, float* U, int cin, int cout, void* stream);
/* the same image straight from an nn.Conv2d weight (Cout,Cin,3,3) with element strides (so, si, sky, skx) — contiguous or
 * channels_last, no permuted copy: mode 0 = forward (Cin -> Cout), mode 1 = input gradient (the convolution dy -> dx with the
 * flipped, transposed weights: Cout -> Cin) */
int crb_winograd2_weights_conv(const float* w, int64_t so, int64_t si, int64_t sky, int64_t skx, float* U, int conv_cin,
                               int conv_cout, int mode, void* stream);
/* the same for n <= 32 weight tensors in ONE launch (the 11 stride-1 layers of the BEV backbone need 22 images per training step, each
 * launch-bound at ~10 us): w[j] (Cout_j, Cin_j, 3, 3) with element strides strides[4 j .. 4 j + 3], U[j] its image for mode[j].
 * w, U, conv_cin, conv_cout, mode, strides are HOST arrays. */
int crb_winograd2_weights_conv_multi(int n, const float* const* w, const int64_t* strides, float* const* U, const int32_t* conv_cin,
                                     const int32_t* conv_cout, const int32_t* mode, void* stream);
int crb_conv3x3_winograd2_nhwc(const float* x, const float* U, float* y, int N, int H, int W, int cin, int cout,
                               const float* bias, int relu, void* stream);
/* training forward that also hands the following BatchNorm its statistics: stats (crb_winograd2_stats_slabs(N,H,W), 2, Cout) f32 =
 * column sums of y and y^2 per slab of outputs (a slab = half of the 64 tiles of one 16 x 4-tile spatial block; outputs outside
 * the map are not counted), every slab written exactly once, no atomics. Feed them to crb_bn_relu_forward_partials(y, N*H*W,
 * Cout, stats, slabs, ...): the BatchNorm's own statistics pass over y (pcdet/models/backbones_2d/base_bev_backbone.py:31-41,
 * nn.BatchNorm2d in training mode) is not launched. No bias, no ReLU (the Conv2d layers of the backbone have neither). */
int64_t crb_winograd2_stats_slabs(int N, int H, int W);
int crb_conv3x3_winograd2_stats_nhwc(const float* x, const float* U, float* y, float* stats, int N, int H, int W, int cin, int cout,
                                     void* stream);

/* a7, round 6: the same convolution with the 16 Winograd GEMMs on the bf16 matrix pipe through an EXACT three-way split of every
 * f32 operand (csrc/winograd_conv4.hip). replaces: the same torch.nn.Conv2d(C, C, 3, padding=1) of
 * pcdet/models/backbones_2d/base_bev_backbone.py:24-41 as crb_conv3x3_winograd2_nhwc. x = x1 + x2 + x3 with x1 = x & 0xffff0000,
 * x2 = (x - x1) & 0xffff0000, x3 = x - x1 - x2 (no rounding: an f32 significand is three bf16 significands); of the nine exact
 * partial products of x * w the six with i + j <= 4 are accumulated in f32 by v_mfma_f32_32x32x16_bf16 (the three dropped ones are
 * below 2^-23 |x w|, one f32 rounding of the product): f32 in, f32 out, errors against an f64 convolution at the level of the f32-MFMA
 * kernel's (tests/test_winograd_gpu.py holds both to the same bars), at 16 / 6 of the f32 MFMA rate. Inf / NaN inputs give NaN (inf -
 * inf in the split), values below 2^-110 lose their low pieces to flushing.
 * Same arguments as the *2 entry points; the weight image is bf16 pieces in LDS order ([Cout/64][Cin/16][xi row][xi][piece][k
 * group][64 rows][8]), crb_winograd4_weights_bytes bytes. crb_winograd4_supported: Cin % 16 == 0, Cout % 64 == 0, H >= 31 (a block of
 * 16 tile rows touches at most two images). One wave per SIMD with all 16 xi of a 32 x 32 block (256 accumulators), raw block shared
 * by the workgroup (halo rows fetched once), four xi-row phases per 16-channel chunk. Stats: crb_winograd4_stats_slabs slabs (one per
 * spatial block and tile half; tile rows per image NOT rounded up to even), same meaning as crb_winograd2_stats_slabs' otherwise. */
int crb_winograd4_supported(int cin, int cout, int H, int W);
int64_t crb_winograd4_weights_bytes(int cin, int cout);
int crb_winograd4_weights_conv(const float* w, int64_t so, int64_t si, int64_t sky, int64_t skx, void* U, int conv_cin, int conv_cout,
                               int mode, void* stream);
int crb_winograd4_weights_conv_multi(int n, const float* const* w, const int64_t* strides, void* const* U, const int32_t* conv_cin,
                                     const int32_t* conv_cout, const int32_t* mode, void* stream);
int crb_conv3x3_winograd4_nhwc(const float* x, const void* U, float* y, int N, int H, int W, int cin, int cout, const float* bias,
                               int relu, void* stream);
int64_t crb_winograd4_stats_slabs(int N, int H, int W);
int crb_conv3x3_winograd4_stats_nhwc(const float* x, const void* U, float* y, float* stats, int N, int H, int W, int cin, int cout,
                                     void* stream);
/* The same convolution with a workgroup tile of 32 tiles x 128 output channels (round 6, third form): V of a spatial block is formed once
 * per 128 output channels instead of once per 64. Cin % 16 == 0, Cout % 128 == 0; weight image: crb_winograd4_weights_conv(_multi) with
 * mode + 2 (same bytes, other order); statistics slabs: crb_winograd4c_stats_slabs (one per block of 8 tile rows x 4 tile columns).
 * Outputs bit-equal to crb_conv3x3_winograd4_nhwc. */
int crb_winograd4c_supported(int cin, int cout, int H, int W);
int crb_conv3x3_winograd4c_nhwc(const float* x, const void* U, float* y, int N, int H, int W, int cin, int cout,
                                const float* bias, int relu, void* stream);
int64_t crb_winograd4c_stats_slabs(int N, int H, int W);
int crb_conv3x3_winograd4c_stats_nhwc(const float* x, const void* U, float* y, float* stats, int N, int H, int W, int cin,
                                      int cout, void* stream);

/* a7 backward: weight gradient of the same convolution in the Winograd domain (csrc/winograd_wgrad.hip):
 * dU[xi][ci][co] = sum over tiles of (B^T d B)[xi][ci] * (A dY A^T)[xi][co] as 16 MFMA GEMMs whose two operands are both
 * produced by transforms inside the kernel, partial sums per range of tiles in the workspace, then dW = G^T dU G added up in
 * range order in double (bit-reproducible).
 * replaces: aten.convolution_backward(..., output_mask = [False, True, False]) = MIOpen's f32 implicit-GEMM wrw kernels for
 *           torch.nn.Conv2d(C, C, 3, padding=1) of pcdet/models/backbones_2d/base_bev_backbone.py:24-41.
 * x (N,H,W,Cin), dy (N,H,W,Cout) f32 NHWC; dw = gradient of the nn.Conv2d weight (Cout,Cin,3,3), written with that tensor's
 * element strides (so, si, sky, skx); Cin % 64 == 0, Cout % 64 == 0. */
int crb_winograd2_wgrad_supported(int cin, int cout, int H, int W);
int64_t crb_winograd2_wgrad_workspace_bytes(int cin, int cout);
int crb_winograd2_wgrad(const float* x, const float* dy, float* dw, int64_t so, int64_t si, int64_t sky, int64_t skx,
                        int N, int H, int W, int cin, int cout, void* workspace, int64_t workspace_bytes, void* stream);

/* Scheduling hint, no reference counterpart: PV-RCNN's keypoint sampling (crb_furthest_point_sampling_stack on a side stream under
 * the backbones) holds one CU per frame for ~5 ms while the persistent one-workgroup-per-CU Winograd launches of the BEV backbone
 * run on the main stream. crb_cu_reservation(cus, stream) right before such a kernel, crb_cu_reservation(0, stream) right after it
 * (same stream: two one-thread launches): the Winograd forward launches in between spread their units over (CUs - cus)
 * workgroups instead of leaving `cus` workgroups queued behind the taken CUs. Only launches that put a workgroup on every CU
 * look at it, and at most half of the CUs are ever announced as taken (a larger `cus` is clamped). Results do not depend on it. */
int crb_cu_reservation(int cus, void* stream);

/* a19: bilinear lookup of the BEV feature map at the keypoints.
 * replaces: VoxelSetAbstraction.interpolate_from_bev_features + bilinear_interpolate_torch
 *           (pcdet/models/backbones_3d/pfe/voxel_set_abstraction.py:176-207, :11-44): per frame four advanced-index gathers of
 *           the (H, W, C) map, four weights from the clamped corner coordinates, a weighted sum; autograd's backward = four
 *           index_put(accumulate) calls.
 * bev (B,H,W,C) f32 NHWC (= the channels_last (B,C,H,W) map), C % 4 == 0; keypoints (M,4) f32 [frame, x, y, z];
 * u = ((x - x_min) * (1 / voxel_x)) * (1 / bev_stride) in f32 (what torch's division of a tensor by a host scalar computes, twice,
 * as the reference writes it), corners floor(u), floor(u) + 1 clamped into the
 * map, weights from the CLAMPED corners, products added left to right: bit-identical to the torch expression. out (M,C).
 * backward: dbev (B,H,W,C) must be ZERO on entry; the four weighted copies of dout are added with float atomics (keypoints that
 * share a cell: sums in arrival order). */
int crb_bev_interpolate_forward(const float* bev, int B, int H, int W, int C, const float* keypoints, int64_t M, float x_min,
                                float y_min, float voxel_x, float voxel_y, float bev_stride, float* out, void* stream);
int crb_bev_interpolate_backward(const float* dout, int B, int H, int W, int C, const float* keypoints, int64_t M, float x_min,
                                 float y_min, float voxel_x, float voxel_y, float bev_stride, float* dbev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CRB_HIP_H */
