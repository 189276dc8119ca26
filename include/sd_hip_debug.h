/* sd_hip_debug.h -- measurement and test switches of libsd_hip.so.  NOT part of the product ABI (include/sd_hip.h): a
 * caller that only includes sd_hip.h can reach no process-wide state besides the event pool and the sd_prof_* accumulators
 * documented there.  The library never reads the environment; every A/B switch the benchmarks (tests/bench_*.py) and the
 * forced-variant parity tests use goes through sd_debug_set.  Process-wide, not thread-safe, meant to be set while no
 * launch is in flight.
 *
 *   key                        default  meaning
 *   gemm.force_bm / force_nst  0 / 0    force tile rows (64|128|256) and LDS ring depth (2..4; 9 = the staggered 256x128
 *                                       family) of every later sd_gemm_* call; 0 / 0 = the dispatch heuristic
 *   gemm.fwd_bump              0        tiles of a forward that shares the GPU with a second stream (SD_FWD_CONCURRENT,
 *                                       sd_hip.h): 0 = follow the caller's flag (then bits 0|1); -1 = never; > 0 = force
 *                                       these bits on every call, flagged or not -- bit0: forward + residual GEMMs 64 -> 128
 *                                       rows, bit1: 128 -> 256, bit2 / bit3: the same for the dX GEMMs, bit4: 64 -> 256,
 *                                       bit5 / bit6: the 128-row tile of bit0 / of the teacher with a 2-stage ring (two
 *                                       workgroups per CU); bits 2-6 were measured and lost
 *   gemm.splitk_max            0        (measurement) most K slices the split-K plan may choose (0 = 8)
 *   gemm.fwd_cu_budget         0        (measurement) workgroups of the persistent 256x128 FORWARD launches of a
 *                                       SD_FWD_CONCURRENT pass (teacher / student gate|up); 0 = one per CU
 *   gemm.persist_balance       0        (measurement) the persistent forward kernel starts as many workgroups as give each the
 *                                       same number of tiles (student gate|up: 192 x 2 tiles instead of 256 with 1 or 2)
 *   gemm.checked_staging       0        pointer staging with a zero page instead of buffer descriptors
 *   gemm.p256_unpaired         0        gemm_p256_kernel with 32-deep half-line stages (round-2 form)
 *   gemm.cu_budget             0        workgroups of the backward's persistent weight-gradient launches; 0 = three quarters
 *                                       of the CUs when they run on the side stream beside the dX chain, else one per CU;
 *                                       -1 = one per CU always.
 *                                       A multi-GPU run where RCCL's kernels hold CUs beside the backward may set 240
 *                                       (speech_distill_amd/ddp.py forwards SD_GEMM_CU_BUDGET)
 *   gemm.no_persist / no_p256  0 / 0    never dispatch the persistent 256x128 / 256x256 kernels
 *   gemm.p256_min_tiles        1024     least number of 256x256 tiles for gemm_p256_kernel
 *   gemm.group_m               0        row tiles per XCD block of the tile order (0 = 4 for 256-row tiles, else 8)
 *   gemm.tn_stag_min           1024     least number of 256x128 tiles for the staggered kernels on a TN (dW) GEMM
 *   gemm.splitk_min_kt         96       least K/64 for sd_gemm_splitk_plan to split
 *   gemm.splitk_min_slice      24       least K/64 per slice (clamped to >= 1)
 *   gemm.no_table              0        ignore the measured shape -> variant table (csrc/sd_gemm_table.inc): heuristic only
 *   model.fuse_student_swiglu  0        SwiGLU in the gate|up GEMM epilogue also when gate|up is kept for the backward
 *   model.shared_layers / shared_layers_train   -1 / -1
 *                                       decoder layers of a SD_FWD_CONCURRENT forward that run with the shared-GPU tiles, the
 *                                       rest run as if alone (-1 = all): for an INFERENCE forward (the teacher) / for a forward
 *                                       that saves activations (the student)
 *   model.overlap_mask         31       sd_qwen3_backward: bit0 lm_head dW, bit1 gain reduces, bit2 attention dQ on the side
 *                                       stream, bit3 grouped per-layer dW, bit4 one batched gain reduce per layer
 *   topk.nt                    0        threads per row of topk_kernel (256|512|1024; 0 = 512)
 *   qk_bwd.blocks              512      workgroups of qknorm_rope_bwd_kernel
 *   attn.variant               0        bit0: forward without the in-wave software pipeline at any T; bit1: with it at any T
 *   reset                      -        restore every default
 */
#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* SD_OK, or SD_ERR_UNSUPPORTED for an unknown key */
int sd_debug_set(const char* key, int64_t value);
/* current value, INT64_MIN for an unknown key */
int64_t sd_debug_get(const char* key);
/* newline-separated list of the keys; returns the bytes needed (cap too small: nothing written) */
int sd_debug_keys(char* buf, int cap);

#ifdef __cplusplus
}
#endif
