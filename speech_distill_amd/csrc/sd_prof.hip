#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <string>
#include <vector>
#include "sd_prof.h"

bool sd_prof_enabled = false;

namespace {
struct Rec { int kind; double work; hipEvent_t a, b; std::string sym; };
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_pool_next = 0;
thread_local int g_last = -1;
struct SymAgg { double ms = 0, work = 0; long count = 0; int kind = 0; };
std::map<std::string, SymAgg> g_syms;  // filled by sd_prof_end, read by sd_prof_symbols

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
  Rec r{kind, work, get_event(), get_event(), std::string()};
  (void)hipEventRecord(r.a, st);
  g_recs.push_back(r);
  *slot = g_last = (int)g_recs.size() - 1;
}
void sd_prof_close(int slot, hipStream_t st) { (void)hipEventRecord(g_recs[slot].b, st); }

void sd_prof_label(const char* fmt, ...) {
  if (g_last < 0 || g_last >= (int)g_recs.size()) return;
  char buf[160];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_recs[g_last].sym = buf;
}

extern "C" int sd_prof_begin(void) {
  g_recs.clear();
  g_syms.clear();
  g_pool_next = 0;
  g_last = -1;
  sd_prof_enabled = true;
  return 0;
}

// Synchronises the device, sums elapsed time / work / launch count per kernel kind (and per kernel symbol, for
// sd_prof_symbols), stops profiling.
extern "C" int sd_prof_end(double* ms, double* work, int64_t* count, int n_kinds) {
  sd_prof_enabled = false;
  if (hipDeviceSynchronize() != hipSuccess) return SD_ERR_WORKSPACE;
  for (int k = 0; k < n_kinds; ++k) { ms[k] = 0; work[k] = 0; count[k] = 0; }
  static const char* kind_names[SD_K_COUNT] = {"gemm_nt", "gemm_nn", "gemm_tn", "attn_fwd_kernel", "attn_bwd_dkv_kernel",
                                               "attn_bwd_dq_kernel", "kd_fwd_kernel", "kd_bwd_kernel", "topk_kernel",
                                               "rmsnorm", "qknorm_rope", "swiglu", "embedding", "optim", "misc",
                                               "gemm_nt_stag"};
  for (const Rec& r : g_recs) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    if (r.kind < n_kinds) { ms[r.kind] += t; work[r.kind] += r.work; count[r.kind] += 1; }
    SymAgg& a = g_syms[r.sym.empty() ? std::string(r.kind < SD_K_COUNT ? kind_names[r.kind] : "?") : r.sym];
    a.ms += t; a.work += r.work; a.count += 1; a.kind = r.kind;
  }
  g_recs.clear();
  g_pool_next = 0;
  g_last = -1;
  return 0;
}

// Per-symbol table of the last sd_prof_end as text, one line per symbol: "symbol\tkind\tms\twork\tlaunches\n".
// Returns the number of bytes needed (call with cap = 0 to size the buffer).
extern "C" int64_t sd_prof_symbols(char* buf, int64_t cap) {
  std::string out;
  char line[320];
  for (const auto& kv : g_syms) {
    snprintf(line, sizeof line, "%s\t%d\t%.6f\t%.6e\t%ld\n", kv.first.c_str(), kv.second.kind, kv.second.ms, kv.second.work,
             kv.second.count);
    out += line;
  }
  if (buf && cap > 0) {
    const size_t n = out.size() < (size_t)(cap - 1) ? out.size() : (size_t)(cap - 1);
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return (int64_t)out.size() + 1;
}
