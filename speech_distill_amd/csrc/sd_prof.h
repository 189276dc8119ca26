// Optional in-library timing of every launch with HIP events recorded on the launch stream (used by
// bench.py to measure per-kernel duration live, inside the timed region).  Off by default; when off a
// scope costs one predictable branch.  This is the only process-wide state in the library.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/sd_hip.h"

void sd_prof_open(int kind, double work, hipStream_t st, int* slot);
void sd_prof_close(int slot, hipStream_t st);
// Names the kernel symbol of the most recently opened record of this thread (printf-style; only called when profiling
// is on).  Launchers that pick among kernel templates call it next to the launch, so that bench.py can report the
// ONE dominant symbol as rocprofv3 prints it (e.g. "gemm_pstag_kernel<4, false, false, 3>").
void sd_prof_label(const char* fmt, ...);
extern bool sd_prof_enabled;
#define SD_PROF_LABEL(...) do { if (sd_prof_enabled) sd_prof_label(__VA_ARGS__); } while (0)

struct SdProfScope {
  int slot = -1;
  hipStream_t st;
  SdProfScope(int kind, double work, hipStream_t s) : st(s) {
    if (sd_prof_enabled) sd_prof_open(kind, work, s, &slot);
  }
  ~SdProfScope() {
    if (slot >= 0) sd_prof_close(slot, st);
  }
};
