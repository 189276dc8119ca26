// sd_debug_set / sd_debug_get (include/sd_hip_debug.h): the ONE door to the library's measurement switches.
#include <stdint.h>
#include <string.h>
#include "../../include/sd_hip.h"
#include "../../include/sd_hip_debug.h"
#include "sd_debug.h"

SdDebug g_sd_debug;
thread_local int t_sd_shared_gpu = 0;

namespace {
struct Key { const char* name; int SdDebug::*field; };
const Key kKeys[] = {
    {"gemm.force_bm", &SdDebug::gemm_force_bm},
    {"gemm.force_nst", &SdDebug::gemm_force_nst},
    {"gemm.checked_staging", &SdDebug::gemm_checked_staging},
    {"gemm.p256_unpaired", &SdDebug::gemm_p256_unpaired},
    {"gemm.cu_budget", &SdDebug::gemm_cu_budget},
    {"gemm.no_persist", &SdDebug::gemm_no_persist},
    {"gemm.no_p256", &SdDebug::gemm_no_p256},
    {"gemm.p256_min_tiles", &SdDebug::gemm_p256_min_tiles},
    {"gemm.group_m", &SdDebug::gemm_group_m},
    {"gemm.tn_stag_min", &SdDebug::gemm_tn_stag_min},
    {"gemm.splitk_min_kt", &SdDebug::gemm_splitk_min_kt},
    {"gemm.splitk_min_slice", &SdDebug::gemm_splitk_min_slice},
    {"gemm.splitk_max", &SdDebug::gemm_splitk_max},
    {"gemm.no_table", &SdDebug::gemm_no_table},
    {"gemm.persist_balance", &SdDebug::gemm_persist_balance},
    {"gemm.fwd_cu_budget", &SdDebug::gemm_fwd_cu_budget},
    {"gemm.fwd_bump", &SdDebug::gemm_fwd_bump},
    {"model.fuse_student_swiglu", &SdDebug::model_fuse_student_swiglu},
    {"model.overlap_mask", &SdDebug::model_overlap_mask},
    {"model.shared_layers", &SdDebug::model_shared_layers},
    {"model.shared_layers_train", &SdDebug::model_shared_layers_train},
    {"topk.nt", &SdDebug::topk_nt},
    {"qk_bwd.blocks", &SdDebug::qk_bwd_blocks},
    {"attn.variant", &SdDebug::attn_variant},
};
}  // namespace

extern "C" int sd_debug_set(const char* key, int64_t value) {
  if (!key) return SD_ERR_SHAPE;
  if (!strcmp(key, "reset")) { g_sd_debug = SdDebug{}; return SD_OK; }
  for (const Key& k : kKeys)
    if (!strcmp(key, k.name)) {
      if (k.field == &SdDebug::gemm_splitk_min_slice && value < 1) value = 1;  // a divisor
      g_sd_debug.*(k.field) = (int)value;
      return SD_OK;
    }
  return SD_ERR_UNSUPPORTED;
}

extern "C" int64_t sd_debug_get(const char* key) {
  if (key)
    for (const Key& k : kKeys)
      if (!strcmp(key, k.name)) return g_sd_debug.*(k.field);
  return INT64_MIN;
}

extern "C" int sd_debug_keys(char* buf, int cap) {
  int need = 0;
  for (const Key& k : kKeys) need += (int)strlen(k.name) + 1;
  if (buf && cap >= need + 1) {
    char* p = buf;
    for (const Key& k : kKeys) { size_t n = strlen(k.name); memcpy(p, k.name, n); p[n] = '\n'; p += n + 1; }
    *p = 0;
  }
  return need + 1;
}
