// Measurement only (bench.py --experiment-cu-hog): N workgroups that hold a CU slot each (4 waves,
// 128 registers per lane: what a communication library's long-running reduction kernel occupies) for a given wall time
// and do nothing else -- the stand-in for RCCL's kernels beside the backward on a box with one GPU.  Bounded by the
// 100 MHz real-time counter: every wave leaves after `us` microseconds whatever happens.
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void cu_hog_kernel(unsigned long long ticks) {
  asm volatile("v_mov_b32 v127, 0" ::: "v127");  // claim 128 VGPRs
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

extern "C" int cu_hog(int n_wg, int us, void* stream) {
  if (n_wg <= 0 || us <= 0 || us > 200000) return -1;
  hipLaunchKernelGGL(cu_hog_kernel, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, (unsigned long long)us * 100ull);
  return (int)hipGetLastError();
}
