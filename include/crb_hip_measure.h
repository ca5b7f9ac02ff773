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
/* measurement builds of the second Winograd design (wrong results): 1 = no MFMAs, 2 = no input transform, 3 = no LDS-DMA in the loop; 4, 5 below */
int crb_winograd2_set_mode(int mode);
/* mode 4 (correct results + stamps) writes 16 uint64 per workgroup {s_memtime: start, after prologue, after chunks, end; wall_clock64 (100 MHz): start, end; XCC id; -}; mode 5 = barrier after the last pair's MFMAs (A/B). NULL = off */
int crb_winograd2_set_debug(void* dev_buf_u64x16_per_wg);
/* A/B: 1 = persistent workgroups (one per CU, contiguous unit ranges, one pipeline), 0 = one unit per workgroup (default) */
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

#ifdef __cplusplus
}
#endif
#endif
