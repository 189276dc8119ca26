// Timing-disabled hipEvent_t sets leased per call and per device from a small process-wide pool (the only process state of
// the stream-ordering helpers): two concurrent calls -- different host threads, different devices -- never share an event,
// and nothing is created on the steady-state path.  An event may be re-recorded by a later call while a wait enqueued by
// an earlier one is still pending: hipStreamWaitEvent captures the record that was current when it was issued.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <vector>

constexpr int kSdEventsPerSet = 12;
struct SdEventSet {
  hipEvent_t ev[kSdEventsPerSet];
  int device;
};
SdEventSet* sd_lease_events();
void sd_return_events(SdEventSet* s);
struct SdEventLease {
  SdEventSet* set = nullptr;
  ~SdEventLease() { if (set) sd_return_events(set); }
};
