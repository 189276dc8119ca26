#include <vector>
#include "sd_prof.h"

bool sd_prof_enabled = false;

namespace {
struct Rec { int kind; double work; hipEvent_t a, b; };
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_pool_next = 0;

hipEvent_t get_event() {
  if (g_pool_next < g_pool.size()) return g_pool[g_pool_next++];
  hipEvent_t e;
  (void)hipEventCreate(&e);
  g_pool.push_back(e);
  ++g_pool_next;
  return e;
}
}  // namespace

void sd_prof_open(int kind, double work, hipStream_t st, int* slot) {
  Rec r{kind, work, get_event(), get_event()};
  (void)hipEventRecord(r.a, st);
  g_recs.push_back(r);
  *slot = (int)g_recs.size() - 1;
}
void sd_prof_close(int slot, hipStream_t st) { (void)hipEventRecord(g_recs[slot].b, st); }

extern "C" int sd_prof_begin(void) {
  g_recs.clear();
  g_pool_next = 0;
  sd_prof_enabled = true;
  return 0;
}

// Synchronises the device, sums elapsed time / work / launch count per kernel kind, stops profiling.
extern "C" int sd_prof_end(double* ms, double* work, int64_t* count, int n_kinds) {
  sd_prof_enabled = false;
  if (hipDeviceSynchronize() != hipSuccess) return SD_ERR_WORKSPACE;
  for (int k = 0; k < n_kinds; ++k) { ms[k] = 0; work[k] = 0; count[k] = 0; }
  for (const Rec& r : g_recs) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    if (r.kind < n_kinds) { ms[r.kind] += t; work[r.kind] += r.work; count[r.kind] += 1; }
  }
  g_recs.clear();
  g_pool_next = 0;
  return 0;
}
