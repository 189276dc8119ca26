// Optional in-library timing of every launch with HIP events recorded on the launch stream (used by
// bench.py to measure per-kernel duration live, inside the timed region).  Off by default; when off a
// scope costs one predictable branch.  This is the only process-wide state in the library.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/sd_hip.h"

void sd_prof_open(int kind, double work, hipStream_t st, int* slot);
void sd_prof_close(int slot, hipStream_t st);
extern bool sd_prof_enabled;

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
