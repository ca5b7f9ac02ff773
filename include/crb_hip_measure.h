/* crb_hip_measure.h — measurement-only entry points of libcrbhip_measure.so (csrc compiled with -DCRB_MEASURE).
 *
 * NOT part of the product boundary: libcrbhip.so exports none of these (tests/test_abi.py asserts it). They select kernel
 * variants for A/B runs, switch on per-workgroup timelines / cycle accounting, or build kernels that SKIP work and return
 * wrong results by design (to time the remaining part). State is process-global. Only tools/ load this library
 * (CRB_MEASURE_LIB=1 makes crbhip._lib load it and parse this header on top of crb_hip.h). No reference counterpart. */
#ifndef CRB_HIP_MEASURE_H
#define CRB_HIP_MEASURE_H
#include "crb_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* A/B of the sort key of crb_mask_sort_chunks: 2 (default) = mask bits re-ranked by frequency inside the chunk, 1 = by the
 * geometry of a 3x3x3 kernel (corners, edges, faces, centre), 0 = numeric mask order */
int crb_mask_sort_set_rank_bits(int mode);
/* measurement builds of the Winograd convolution (wrong results): 1 = no MFMAs, 2 = no staging of the next chunk */
int crb_winograd_set_mode(int mode);
/* measurement builds of the second Winograd design (wrong results): 1 = no MFMAs, 2 = no input transform, 3 = no LDS-DMA in the loop; 5 .. 13: see csrc/winograd_conv2.hip; 4 below */
int crb_winograd2_set_mode(int mode);
/* measurement builds of the split-bf16 Winograd kernel (wrong results): 1 = no MFMAs, 2 = no input transform, 3 = no LDS-DMA in the loop,
 * 4 = no operand reads, 5 = no U copies, 6 = no raw copies, 7 = no V stores, 8 = no output stores, 9 = stamps (csrc/winograd_conv4.hip) */
int crb_winograd4_set_mode(int mode);
/* A/B: 1 = the product kernel (one 512-register wave per SIMD), 2 = the second form (two waves per SIMD, xi split over wave pairs, pipelined) */
int crb_winograd4_set_variant(int v);
/* A/B of the farthest-point sampling kernel for the in-register sizes: 2 = fps2_kernel (packed updates, DPP reductions, one barrier per
 * round; default), 1 = fps_kernel (round 2) */
int crb_fps_set_variant(int v);
/* mode 9 (correct results + s_memtime sums per (workgroup, wave): {counter wait, barrier, phase head, phase body, epilogue, total, chunks, units}): 8 uint64 per wave, 8 waves per workgroup; NULL = off */
int crb_winograd4_set_debug(void* dev_buf_u64x64_per_wg);
/* mode 4 (correct results + stamps) writes 16 uint64 per workgroup {s_memtime: start, after prologue, after chunks, cycles parked at the chunk barriers; wall_clock64 (100 MHz): start, end; XCC id; units; per-stage cycle sums}. NULL = off */
int crb_winograd2_set_debug(void* dev_buf_u64x16_per_wg);
/* A/B: 1 = persistent workgroups (one per CU, contiguous unit ranges, one pipeline; default), 0 = one unit per workgroup */
int crb_winograd2_set_persistent(int on);
/* measurement builds of the Winograd weight gradient (wrong results): 1 = no MFMAs, 2 = no transforms, 3 = no DMA / gradient loads in the loop */
int crb_winograd2_wgrad_set_mode(int mode);
/* measurement builds of crb_tables_finish's chunk pass (wrong tables by design): bit 0 = no sort, bit 1 = no packed-index fill,
 * bit 2 = no pair lists */
int crb_tables_set_skip(int bits);
/* measurement knob: 16-row tiles per wave (1 or 2; 0 = default) */
int crb_sparse_conv_bf16x3_set_tiles_per_wave(int tpw);
/* measurement builds of the 64x64 kernel (wrong results): 1 = no MFMAs, 2 = no row gathers, 3 = no W hand-over, 4 = 2+3 */
int crb_sparse_conv_bf16x3_set_mode(int mode);
/* kernel-variant knob for A/B measurements only: 0 = default (v2 kernel where Cin,Cout are multiples of 16 and Cin <= 64,
 * else v1); 1|2|4 = v1 with 64*subt rows per workgroup; 8 = v2. Results are identical up to f32 summation order. */
int crb_sparse_conv_set_subtiles(int subt);
/* A/B: 1 = row-contiguous gathers + in-quad DPP transpose in the compact-table kernel at Cin = 64 (same products, another
 * grouping of the channels over the MFMA steps: equal to f32 rounding, not bit-equal) */
int crb_sparse_conv_set_rowc(int on);
/* low-channel forward kernel (C_in in {4,16}, C_out = 16): 1 = product default (C_in = 4 only), 0 = never, 2 = both shapes,
 * >= 16 = both shapes with the weights resident in LDS and that many 16-wave workgroups (A/B) */
int crb_sparse_conv_set_lowchannel(int v);
/* measurement builds: after launches under crb_sparse_conv_set_subtiles(32) (64x64 kernel with s_memtime accounting), copy the
 * 16 accumulated counters to host memory and clear them: [0] waves [1] total cycles [2] prologue [3] load issue [4] MFMA
 * block [5] W store [6] barrier wait [7] epilogue [8] phases [9] phases with MFMA work [10] W fetch issue [11] row-index
 * LDS read [12] wait for the previous phase's gather prefetch. Synchronises the device. */
int crb_sparse_conv_timing(uint64_t* out16_host);
int crb_sparse_conv_set_wgrad_splits(int splits);    /* measurement knob: workgroups per offset (multiple of 8), 0 = default */
int crb_sparse_conv_set_wgrad_debug(void* dev_buf_u64x4_per_wg); /* measurement runs: per-workgroup {start, end, HW_ID, XCC_ID | steps<<32} of the v2 wgrad; NULL = off */
int crb_sparse_conv_set_wgrad_mode(int mode);        /* measurement builds of the 64x64 wgrad: 1 = no MFMAs, 2 = no gather pipeline (results are wrong by design), 0 = normal */
int crb_sparse_conv_set_wgrad_v1(int on);            /* measurement knob: 1 = the v1 (16x16x4, register-gather) wgrad kernel for every shape */
/* traversal order of the BatchNorm passes: bit 0 = statistics passes from the last row range to the first, bit 1 = apply passes
 * (results unchanged; A/B of how much of the second read of a tensor the 256 MB Infinity Cache serves) */
int crb_bn_set_order(int bits);

/* ---- kernels that left the product library in round 5 (dead weight there: VERDICT r04 item 8) ---- */

/* OPT-IN arithmetic contract "bf16x3" for the same gather-GEMM (exact f32 above stays the default): every operand is split
 * into two bf16 values, x = x_hi + x_lo (+ a residual <= 2^-18 |x|), and a product is taken as x_lo*w_hi + x_hi*w_lo +
 * x_hi*w_hi on the bf16 MFMA (products exact, f32 accumulation; x_lo*w_lo dropped). Stated bound, checked by
 * tests/test_spconv_gpu.py: |y - y_exact| <= 2^-16 * sum |x||w| over the gathered products of the output element (plus
 * f32 accumulation error). bf16 keeps the f32 exponent range: no scaling, no overflow case of its own. Model-level reading
 * (tests/test_second_gpu.py): a SECOND training step reproduces the f32 loss to 2e-6 and the dense-head gradients to 4e-5,
 * the weight gradients of the sparse backbone — sums that cancel to ~1e-3 of their terms — to 1-4 % of their largest entry. Needs the compact
 * table of crb_nbr_compact; workspace >= crb_sparse_conv_bf16x3_workspace_bytes (holds the split copy of W). n_in = rows of
 * X (the gathers are bounds-checked buffer loads; n_in*cin*4 must stay below 2^31).
 * No reference counterpart: spconv-cu113 v2.1.21 multiplies in f32 (or fp16 under AMP, which the reference does not use). */
int crb_sparse_conv_bf16x3_supported(int cin, int cout);
int64_t crb_sparse_conv_bf16x3_workspace_bytes(int K, int cin, int cout);
int crb_sparse_conv_forward_bf16x3(const float* X, const float* W, const uint32_t* cmask, const int32_t* cbase,
                                   const int32_t* packed, const int32_t* perm, const int32_t* tile_order, float* Y,
                                   int64_t n_in, int64_t n_out, int K, int cin, int cout, void* workspace,
                                   int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a7 (stretch)  3x3 stride-1 pad-1 convolution on channels_last maps as Winograd F(2x2,3x3) on the f32 MFMA
 * replaces: torch.nn.Conv2d(C, C, 3, padding=1) of the BEV backbone (pcdet/models/backbones_2d/base_bev_backbone.py:24-41;
 *           cuDNN in the reference, MIOpen's f32 implicit GEMM here) for the stride-1 layers whose channel counts pass
 *           crb_winograd_supported (Cin % 32 == 0, Cout % 128 == 0). OPT-IN on the Python side: results differ from a direct
 *           convolution by f32 rounding of the transforms (<= 1e-5 of the output scale on unit-scale data, tested).
 * x (N,H,W,Cin) f32 NHWC, y (N,H,W,Cout); weights first through crb_winograd_weights: g (3,3,Cin,Cout) [ky][kx][ci][co]
 * -> U (16,Cin,Cout) (crb_winograd_weights_bytes). The input gradient of the same layer is the same call on dy with
 * g'[ky][kx][co][ci] = w[co][ci][2-ky][2-kx]. bias (Cout) or NULL, relu 0/1: epilogue on the output. */
int crb_winograd_supported(int cin, int cout);
int64_t crb_winograd_weights_bytes(int cin, int cout);
int crb_winograd_weights(const float* g, float* U, int cin, int cout, void* stream);
int crb_conv3x3_winograd_nhwc(const float* x, const float* U, float* y, int N, int H, int W, int cin, int cout,
                              const float* bias, int relu, void* stream);

/* measurement builds of the training set-abstraction passes (wrong results): bit 0 = no MFMAs in the forward GEMM, bit 1 = every P row is
 * row 0 (no gather misses), bit 2 = no first layer */
int crb_sa_mlp2_train_set_skip(int bits);
/* FETCH_SIZE calibration on the Winograd forward kernel's access pattern: LDS-DMA (global_load_lds_dwordx4) reads of `pieces` pieces of
 * 32 bytes, `stride_bytes` apart (32 = dense, 512 = the 8-channel pieces of adjacent pixels of a 128-channel NHWC map); pass_mask bit p
 * = a sweep over the p-th 32-byte piece of every stride (p < 4). Requested bytes = pieces * 32 * popcount(pass_mask). sink256: 256 floats. */
int crb_probe_lds_dma(const float* src, int64_t pieces, int stride_bytes, int pass_mask, float* sink256, void* stream);
/* one workgroup per CU streams an L2-resident image into LDS `iters` times: variant 0 = LDS-DMA, 1 = global_load_dwordx4 + ds_write_b128;
 * threads 256 | 512, depth 4 | 8 instructions in flight per wave (csrc/probe_floor.hip) */
int crb_probe_stream(int variant, int threads, int depth, int cus, const float* src, int64_t bytes, int iters, float* sink, void* stream);
/* latency-floor probes of the low-channel subm layers (csrc/probe_floor.hip): the memory side of the gather chain on 16-channel rows,
 * no weights / MFMA. variant 0 = copy y[i] = x[i]; 1 = two dependent round trips (ell (n,8) fixed-stride neighbour list, -1 = none,
 * rows summed); 2 = three (cmask / cbase -> packed -> rows: the compact table's chain). y (n,16). */
int crb_probe_gather_chain(int variant, const float* x, int64_t n, const uint32_t* cmask, const int32_t* cbase,
                           const int32_t* packed, const int32_t* ell, float* y, void* stream);
/* VERDICT r04 item 6a, measured and NOT adopted (profiles/r05_time_wino_bnbwd.txt, tools/time_wino_bnbwd.py): the BatchNorm
 * backward's reduction pass inside the Winograd input-gradient kernel's epilogue. */
/* the input-gradient launch of layer L+1 when its output dz is the gradient w.r.t. relu(batchnorm_L(bn_y)): the same kernel with an
 * epilogue that also writes the slab sums (crb_winograd2_stats_slabs, 2, Cout) of dz [z > 0] and dz [z > 0] xhat (xhat = (bn_y - mean)
 * invstd, z = gamma xhat + beta; relu = 0: no mask) -> crb_bn_relu_backward_partials: the reduction pass of that BatchNorm's backward
 * over (bn_y, dz) is not launched (base_bev_backbone.py:31-41 in training; autograd's native_batch_norm_backward in the reference).
 * x = dy of layer L+1, U = its input-gradient weight image, y = dz (N,H,W,Cout), bn_y (N,H,W,Cout). */
int crb_conv3x3_winograd2_bnbwd_nhwc(const float* x, const float* U, float* y, float* stats, int N, int H, int W, int cin, int cout,
                                     const float* bn_y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                     int relu, void* stream);
/* crb_bn_relu_backward with the reduction done by the producer of dz: slab_sums (n_slabs, 2, C) = column sums of dz [z > 0] and
 * dz [z > 0] xhat over disjoint sets of rows that cover all n (crb_conv3x3_winograd2_bnbwd_nhwc) */
int crb_bn_relu_backward_partials(const float* x, const float* dz, int64_t dz_row_stride, int64_t n, int C, const float* slab_sums,
                                  int64_t n_slabs, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                  int relu, float* dx, float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes,
                                  int32_t* tickets, void* stream);

#ifdef __cplusplus
}
#endif
#endif
