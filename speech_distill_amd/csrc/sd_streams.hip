// Which HIP streams really run side by side?  The HIP runtime multiplexes streams onto a handful of hardware queues
// (four by default); two streams that land on the same queue execute strictly one after the other.  The step relies on
// three overlaps (the frozen teacher beside the student's forward, the weight-gradient GEMMs beside the dX chain, the
// RCCL all-reduce beside the backward), and which streams alias depends on how many streams the process created
// before ours -- measured: under torch.distributed the teacher's stream shared the main stream's queue and the forward
// phase took 13.0 ms instead of 11.6.  sd_streams_overlap answers the question by experiment so that the host can
// pick streams that do overlap (ops.concurrent_stream).
#include <hip/hip_runtime.h>

#include "../../include/sd_hip.h"

namespace {

// busy-waits on the constant-rate wall clock: ends after `ticks` whatever else the chip is doing
__global__ void spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

}  // namespace

extern "C" int sd_streams_overlap(void* stream_a, void* stream_b, float spin_us, int* overlap) {
  if (!overlap || spin_us <= 0.f || spin_us > 20000.f) return SD_ERR_SHAPE;
  hipStream_t a = (hipStream_t)stream_a, b = (hipStream_t)stream_b;
  *overlap = 0;
  if (a == b) return 0;
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess) return SD_ERR_WORKSPACE;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
  hipEvent_t e[3] = {nullptr, nullptr, nullptr};
  int rc = 0;
  for (int i = 0; i < 3 && !rc; ++i)
    if (hipEventCreate(&e[i]) != hipSuccess) rc = SD_ERR_WORKSPACE;
  // both streams idle first, so that the only thing that can hold b's kernel back is a's
  if (!rc && (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess)) rc = SD_ERR_WORKSPACE;
  if (!rc) {
    const long long ticks = (long long)(spin_us * 1e-3f * (float)khz);
    if (hipEventRecord(e[0], a) != hipSuccess) rc = SD_ERR_WORKSPACE;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, ticks);
    if (hipEventRecord(e[1], a) != hipSuccess) rc = SD_ERR_WORKSPACE;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, (long long)0);
    if (hipEventRecord(e[2], b) != hipSuccess) rc = SD_ERR_WORKSPACE;
    if (hipEventSynchronize(e[1]) != hipSuccess || hipEventSynchronize(e[2]) != hipSuccess) rc = SD_ERR_WORKSPACE;
    float t_a = 0.f, t_b = 0.f;
    if (!rc && (hipEventElapsedTime(&t_a, e[0], e[1]) != hipSuccess || hipEventElapsedTime(&t_b, e[0], e[2]) != hipSuccess))
      rc = SD_ERR_WORKSPACE;
    if (!rc) *overlap = t_b < 0.5f * t_a ? 1 : 0;  // b's kernel finished while a was still spinning
  }
  for (hipEvent_t ev : e)
    if (ev) (void)hipEventDestroy(ev);
  return rc;
}
