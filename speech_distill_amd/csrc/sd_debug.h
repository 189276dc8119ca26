// Measurement / test switches of the library (include/sd_hip_debug.h: sd_debug_set / sd_debug_get).  One plain struct,
// defaults = the product behaviour; nothing in the library reads the environment.
#pragma once

struct SdDebug {
  // sd_gemm.hip
  int gemm_force_bm = 0, gemm_force_nst = 0;  // 0 = heuristic; else tile rows (64|128|256) and ring depth (2..4, 9 = staggered)
  int gemm_checked_staging = 0;               // pointer-based staging with a zero page instead of buffer descriptors
  int gemm_p256_unpaired = 0;                 // gemm_p256_kernel with 32-deep half-line stages
  int gemm_cu_budget = 0;                     // workgroups of the backward's persistent weight-gradient launches (0 = 3/4 of the CUs beside the dX chain, -1 = all)
  int gemm_no_persist = 0, gemm_no_p256 = 0;
  int gemm_p256_min_tiles = 1024;
  int gemm_group_m = 0;
  int gemm_tn_stag_min = 1024;
  int gemm_splitk_min_kt = 96, gemm_splitk_min_slice = 24;
  int gemm_splitk_max = 0;  // (measurement) most K slices the split-K plan may choose (0 = 8)
  int gemm_fwd_cu_budget = 0;    // (measurement) workgroups of the persistent FORWARD launches of a SD_FWD_CONCURRENT pass (0 = one per CU)
  int gemm_persist_balance = 0;  // (measurement) persistent forward launches: as many workgroups as give every one the same number of tiles
  int gemm_no_table = 0;  // ignore the measured shape -> variant table (sd_gemm_table.inc)
  // tiles of a pass that shares the GPU with another stream (SD_FWD_CONCURRENT): 0 = follow the caller's flag (then bits
  // 0|1), -1 = never, > 0 = force these bits on every call: bit0 forward + residual GEMMs 64 -> 128 rows, bit1 128 -> 256,
  // bit2 / bit3 the same for the dX GEMMs, bit4 64 -> 256, bit5 / bit6 2-stage 128-row tiles (measurement)
  int gemm_fwd_bump = 0;
  // sd_model.hip
  int model_fuse_student_swiglu = 0;
  // decoder layers of a SD_FWD_CONCURRENT forward that are run with the shared-GPU tiles (-1 = all): inference / training pass
  int model_shared_layers = -1, model_shared_layers_train = -1;
  int model_overlap_mask = 31;  // bit0 lm_head dW, bit1 gain reduces, bit2 attention dQ, bit3 grouped per-layer dW, bit4 batched gain reduce
  // sd_topk.hip / sd_elementwise.hip / sd_attn.hip
  int topk_nt = 0;
  int qk_bwd_blocks = 512;
  int attn_variant = 0;  // bit0: forward without the in-wave pipeline at any T; bit1: with it at any T
};
extern SdDebug g_sd_debug;

// Set by the model runner while it enqueues a forward that the caller runs BESIDE another pass on a second stream
// (SD_FWD_CONCURRENT, include/sd_hip.h); read by the GEMM dispatch.  Per host thread, like the launches themselves.
extern thread_local int t_sd_shared_gpu;
struct SdSharedGpuScope {
  int prev;
  explicit SdSharedGpuScope(int on) : prev(t_sd_shared_gpu) { t_sd_shared_gpu = on; }
  ~SdSharedGpuScope() { t_sd_shared_gpu = prev; }
};
