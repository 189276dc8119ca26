// Host-side runner: one C call launches the whole Qwen3 decoder forward (student or frozen teacher)
// or the whole student backward on a HIP stream, kernel after kernel, with every activation laid
// out in one caller-provided HBM buffer (288 GB per MI355X: nothing is recomputed, nothing is
// re-allocated).  Mirrors HF Qwen3ForCausalLM.forward (modeling_qwen3.py:381-441, layer 304-323)
// as the reference calls it at train.py:54 (student, with grad) and train.py:60-69 (teacher, no grad).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/sd_hip.h"
#include "sd_events.h"
#include "sd_debug.h"

namespace {

inline int64_t al(int64_t x) { return (x + 255) & ~(int64_t)255; }

struct Sizes {
  int M, h, I, QD, KD, QKV, QK, V, L, Hq, Hkv;
  int64_t x, rstd, qkv, qk, ao, lse, gu, act;
  Sizes(const sd_qwen3_dims* d, int B, int T) {
    M = B * T; h = d->hidden; I = d->inter; Hq = d->n_q; Hkv = d->n_kv;
    QD = Hq * d->head_dim; KD = Hkv * d->head_dim; QKV = QD + 2 * KD; QK = QD + KD; V = d->vocab; L = d->layers;
    x = al((int64_t)M * h * 2); rstd = al((int64_t)M * 4); qkv = al((int64_t)M * QKV * 2); qk = al((int64_t)M * QK * 2);
    ao = al((int64_t)M * QD * 2); lse = al((int64_t)B * Hq * T * 4); gu = al((int64_t)M * 2 * I * 2);
    act = al((int64_t)M * I * 2);
  }
  int64_t per_layer() const { return 4 * x + 2 * rstd + qkv + qk + ao + lse + gu + act; }
  int64_t tail() const { return 3 * x + rstd; }  // x_last, rstd_f, xn_f, xn_rows (head rows gathered)
  int64_t ssq() const { return al((int64_t)M * 16 * 4); }  // one set of per-tile sums of squares (folded norms)
  // activation storage by mode (sd_hip.h SD_SAVE_*): every layer's buffers | L layer inputs + two layer work sets
  // (recompute; two, because the grouped dW of layer l still reads its set while layer l-1 is recomputed) | one set
  // + a ping-pong x (inference)
  int64_t body(int save) const {
    return save == SD_SAVE_ALL ? (int64_t)L * per_layer() : save == SD_SAVE_LAYER_INPUTS ? (int64_t)L * x + 2 * per_layer()
                                                                                          : per_layer() + x;  // both inference modes
  }
};

struct LayerActs {
  char *x_in, *rstd1, *xn1, *qkv, *qk, *ao, *lse, *x_mid, *rstd2, *xn2, *gu, *act;
};

LayerActs carve(const Sizes& s, char* p) {
  LayerActs a;
  a.x_in = p; p += s.x;
  a.rstd1 = p; p += s.rstd;
  a.xn1 = p; p += s.x;
  a.qkv = p; p += s.qkv;
  a.qk = p; p += s.qk;
  a.ao = p; p += s.ao;
  a.lse = p; p += s.lse;
  a.x_mid = p; p += s.x;
  a.rstd2 = p; p += s.rstd;
  a.xn2 = p; p += s.x;
  a.gu = p; p += s.gu;
  a.act = p; p += s.act;
  return a;
}

// The buffers a layer's weight-gradient GEMMs read (dxa = gradient entering the layer, dgu, dxb, dqkv) and the gain
// partials reduced on the side stream exist twice, used by layer parity: the grouped dW launch of layer l runs on
// the side stream under the dX chain of layer l-1, which writes the other set.
struct BwdScratch {
  char *dxa[2], *dxb[2], *dqkv[2], *dgu[2], *ws_norm2[2], *ws_qk[2], *ws_norm1[2];
  char *dxn, *dqk, *dao, *delta, *dact, *ws_norm, *ws_splitk;
  int64_t total, splitk_bytes;
  BwdScratch(const Sizes& s, char* p) {
    char* p0 = p;
    for (int i = 0; i < 2; ++i) {
      dxa[i] = p; p += s.x;
      dxb[i] = p; p += s.x;
      dqkv[i] = p; p += s.qkv;
      dgu[i] = p; p += s.gu;
      ws_norm2[i] = p; p += al(sd_rmsnorm_bwd_workspace_bytes(s.M, s.h));
      ws_qk[i] = p; p += al(sd_qknorm_rope_bwd_workspace_bytes(s.M, s.Hq, s.Hkv));
      ws_norm1[i] = p; p += al(sd_rmsnorm_bwd_workspace_bytes(s.M, s.h));
    }
    dxn = p; p += s.x;
    dqk = p; p += s.qk;
    dao = p; p += s.ao;
    delta = p; p += s.lse;
    dact = p; p += s.act;
    ws_norm = p; p += al(sd_rmsnorm_bwd_workspace_bytes(s.M, s.h));
    splitk_bytes = sd_gemm_splitk_workspace_bytes(s.M, s.h, s.V);
    for (int k : {s.QKV, 2 * s.I, s.QD, s.I}) {
      const int64_t b1 = sd_gemm_splitk_workspace_bytes(s.M, s.h, k);
      splitk_bytes = b1 > splitk_bytes ? b1 : splitk_bytes;
    }
    ws_splitk = p; p += al(splitk_bytes);
    total = p - p0;
  }
};

#define RUN(call) do { int e__ = (call); if (e__) return e__; } while (0)

// where layer l's buffers live in `acts` for a training forward (SD_SAVE_ALL / SD_SAVE_LAYER_INPUTS)
LayerActs layer_acts(const Sizes& s, char* base, int l, int save) {
  if (save == SD_SAVE_ALL) return carve(s, base + (int64_t)l * s.per_layer());
  LayerActs a = carve(s, base + (int64_t)s.L * s.x + (int64_t)(l & 1) * s.per_layer());
  a.x_in = base + (int64_t)l * s.x;
  return a;
}

// One decoder layer (HF modeling_qwen3.py:227-250): a.x_in -> x_out, every intermediate into `a`.  x_out == nullptr
// stops after the SwiGLU (the backward's recompute does not need the layer output again); keep_gu: gate|up is kept
// for the backward.
int layer_forward(const sd_qwen3_dims* d, const Sizes& s, const LayerActs& a, const sd_qwen3_layer& w, char* x_out,
                  bool keep_gu, const int32_t* kv_len, const void* cos_tab, const void* sin_tab, int B, int T,
                  void* stream) {
  const float scale = 0.08838834764831845f;  // 128^-1/2
  RUN(sd_rmsnorm_fwd(a.x_in, w.ln1, a.xn1, (float*)a.rstd1, s.M, s.h, d->eps, stream));
  // q|k|v projection with q/k-norm + RoPE in the GEMM epilogue (one head = one 128-column tile)
  int rc = sd_gemm_qkv_rope(a.xn1, w.wqkv, a.qkv, a.qk, w.q_gain, w.k_gain, cos_tab, sin_tab, s.M, T, s.Hq, s.Hkv, s.h,
                            d->eps, stream);
  if (rc == SD_ERR_UNSUPPORTED) {
    RUN(sd_gemm_bf16(a.xn1, w.wqkv, a.qkv, nullptr, s.M, s.QKV, s.h, s.h, s.h, s.QKV, 0, 0, 0, stream));
    RUN(sd_qknorm_rope_fwd(a.qkv, w.q_gain, w.k_gain, cos_tab, sin_tab, a.qk, s.M, T, s.Hq, s.Hkv, d->eps, stream));
  } else if (rc) {
    return rc;
  }
  RUN(sd_attn_fwd(a.qk, a.qk + (int64_t)s.QD * 2, a.qkv + (int64_t)(s.QD + s.KD) * 2, a.ao, (float*)a.lse, kv_len,
                  s.QK, s.QK, s.QKV, s.QD, B, T, s.Hq, s.Hkv, 128, scale, stream));
  RUN(sd_gemm_bf16(a.ao, w.wo, a.x_mid, a.x_in, s.M, s.h, s.QD, s.QD, s.QD, s.h, s.h, 0, 0, stream));
  RUN(sd_rmsnorm_fwd(a.x_mid, w.ln2, a.xn2, (float*)a.rstd2, s.M, s.h, d->eps, stream));
  // gate|up projection: SwiGLU runs in the GEMM epilogue when gate|up need not be kept (no backward follows:
  // the frozen teacher).  With the 2*I-wide store as well the fused epilogue is no faster than the separate
  // elementwise pass (tests/bench_fused.py), so the student keeps the two-kernel form.
  rc = (keep_gu && !g_sd_debug.model_fuse_student_swiglu)
           ? SD_ERR_UNSUPPORTED
           : sd_gemm_swiglu(a.xn2, w.wgu, keep_gu ? a.gu : nullptr, a.act, s.M, s.I, s.h, stream);
  if (rc == SD_ERR_UNSUPPORTED) {
    RUN(sd_gemm_bf16(a.xn2, w.wgu, a.gu, nullptr, s.M, 2 * s.I, s.h, s.h, s.h, 2 * s.I, 0, 0, 0, stream));
    RUN(sd_swiglu_fwd(a.gu, a.act, s.M, s.I, stream));
  } else if (rc) {
    return rc;
  }
  if (!x_out) return 0;
  RUN(sd_gemm_bf16(a.act, w.wdown, x_out, a.x_mid, s.M, s.h, s.I, s.I, s.I, s.h, s.h, 0, 0, stream));
  return 0;
}

// The same layer for the frozen teacher with the two RMSNorm gains folded into w.wqkv / w.wgu (SD_SAVE_NONE_FOLDED,
// HF:59-64 + 252-254 / 81-83): no norm launch, no normalised copy of the row.  ssq_in: partial sums of squares of a.x_in
// (from the embedding or the previous layer's down projection); ssq_mid: scratch for those of a.x_mid; ssq_next
// (nullable): where the down projection leaves those of x_out for the next layer.  w.ln1 / w.ln2 are not read.
int layer_forward_folded(const sd_qwen3_dims* d, const Sizes& s, const LayerActs& a, const sd_qwen3_layer& w, char* x_out,
                         const int32_t* kv_len, const void* cos_tab, const void* sin_tab, int B, int T, const float* ssq_in,
                         float* ssq_mid, float* ssq_next, void* stream) {
  const float scale = 0.08838834764831845f;
  RUN(sd_gemm_qkv_rope_rs(a.x_in, w.wqkv, a.qkv, a.qk, w.q_gain, w.k_gain, cos_tab, sin_tab, ssq_in, s.M, T, s.Hq, s.Hkv,
                          s.h, d->eps, stream));
  RUN(sd_attn_fwd(a.qk, a.qk + (int64_t)s.QD * 2, a.qkv + (int64_t)(s.QD + s.KD) * 2, a.ao, (float*)a.lse, kv_len,
                  s.QK, s.QK, s.QKV, s.QD, B, T, s.Hq, s.Hkv, 128, scale, stream));
  RUN(sd_gemm_bf16_ssq(a.ao, w.wo, a.x_mid, a.x_in, ssq_mid, s.M, s.h, s.QD, s.QD, s.QD, s.h, s.h, stream));
  RUN(sd_gemm_swiglu_rs(a.x_mid, w.wgu, nullptr, a.act, ssq_mid, d->eps, s.M, s.I, s.h, stream));
  if (ssq_next) RUN(sd_gemm_bf16_ssq(a.act, w.wdown, x_out, a.x_mid, ssq_next, s.M, s.h, s.I, s.I, s.I, s.h, s.h, stream));
  else RUN(sd_gemm_bf16(a.act, w.wdown, x_out, a.x_mid, s.M, s.h, s.I, s.I, s.I, s.h, s.h, 0, 0, stream));
  return 0;
}

bool fold_supported(const sd_qwen3_dims* d) {
  const int h = d->hidden;
  return d->head_dim == 128 && (h % 512) == 0 && h / 128 <= 16 && (d->inter % 64) == 0;
}

}  // namespace

namespace {
std::mutex g_ev_mu;
std::vector<SdEventSet*> g_ev_free;
}  // namespace

SdEventSet* sd_lease_events() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  {
    std::lock_guard<std::mutex> lk(g_ev_mu);
    for (size_t i = 0; i < g_ev_free.size(); ++i)
      if (g_ev_free[i]->device == dev) {
        SdEventSet* s = g_ev_free[i];
        g_ev_free.erase(g_ev_free.begin() + i);
        return s;
      }
  }
  SdEventSet* s = new SdEventSet;
  s->device = dev;
  for (int i = 0; i < kSdEventsPerSet; ++i)
    if (hipEventCreateWithFlags(&s->ev[i], hipEventDisableTiming) != hipSuccess) {
      for (int j = 0; j < i; ++j) (void)hipEventDestroy(s->ev[j]);
      delete s;
      return nullptr;
    }
  return s;
}
void sd_return_events(SdEventSet* s) {
  std::lock_guard<std::mutex> lk(g_ev_mu);
  g_ev_free.push_back(s);
}

extern "C" int sd_abi_version(void) { return 1; }

extern "C" int sd_qwen3_fold_supported(const sd_qwen3_dims* d) { return d && fold_supported(d) ? 1 : 0; }

extern "C" int64_t sd_qwen3_acts_bytes(const sd_qwen3_dims* d, int B, int T, int save) {
  Sizes s(d, B, T);
  save &= ~SD_FWD_CONCURRENT;
  if (save < SD_SAVE_NONE || save > SD_SAVE_NONE_FOLDED) return SD_ERR_SHAPE;
  if (save == SD_SAVE_NONE_FOLDED) return fold_supported(d) ? s.body(SD_SAVE_NONE) + s.tail() + 2 * s.ssq() : SD_ERR_UNSUPPORTED;
  return s.body(save) + s.tail();
}

extern "C" int64_t sd_qwen3_bwd_scratch_bytes(const sd_qwen3_dims* d, int B, int T) {
  Sizes s(d, B, T);
  BwdScratch b(s, nullptr);
  return b.total;
}

extern "C" int sd_qwen3_forward(const sd_qwen3_dims* d, const sd_qwen3_params* p, const int64_t* ids,
                                const int32_t* kv_len, const void* cos_tab, const void* sin_tab, void* acts,
                                int64_t acts_bytes, void* logits, int B, int T, int save, void* stream) {
  return sd_qwen3_forward_rows(d, p, ids, kv_len, cos_tab, sin_tab, acts, acts_bytes, logits, nullptr, 0, B, T, save,
                               stream);
}

extern "C" int sd_qwen3_forward_rows(const sd_qwen3_dims* d, const sd_qwen3_params* p, const int64_t* ids,
                                     const int32_t* kv_len, const void* cos_tab, const void* sin_tab, void* acts,
                                     int64_t acts_bytes, void* logits, const int64_t* head_rows, int n_head_rows, int B,
                                     int T, int save, void* stream) {
  if (d->head_dim != 128) return SD_ERR_UNSUPPORTED;
  if (B <= 0 || T <= 0) return SD_ERR_SHAPE;
  if (head_rows && (n_head_rows <= 0 || n_head_rows > B * T)) return SD_ERR_SHAPE;
  Sizes s(d, B, T);
  const bool concurrent = (save & SD_FWD_CONCURRENT) != 0;
  save &= ~SD_FWD_CONCURRENT;
  if (save < SD_SAVE_NONE || save > SD_SAVE_NONE_FOLDED) return SD_ERR_SHAPE;
  const bool folded = save == SD_SAVE_NONE_FOLDED;
  if (folded) {
    if (!fold_supported(d)) return SD_ERR_UNSUPPORTED;
    if (acts_bytes < sd_qwen3_acts_bytes(d, B, T, save)) return SD_ERR_WORKSPACE;
    save = SD_SAVE_NONE;  // same buffers as the plain inference forward, plus the two ssq sets behind the tail
  }
  if (acts_bytes < sd_qwen3_acts_bytes(d, B, T, save)) return SD_ERR_WORKSPACE;
  char* base = (char*)acts;
  char* tail = base + s.body(save);
  char* x_last = tail;
  char* rstd_f = tail + s.x;
  char* xn_f = rstd_f + s.rstd;
  char* xn_rows = xn_f + s.x;
  char* pong = base + s.per_layer();  // inference only
  float* ssq_a = (float*)(tail + s.tail());  // folded inference only
  float* ssq_b = (float*)(tail + s.tail() + s.ssq());

  char* x_cur = save ? layer_acts(s, base, 0, save).x_in : base;
  if (folded) RUN(sd_embedding_fwd_ssq(ids, p->embed, x_cur, ssq_a, s.M, s.h, s.V, stream));
  else RUN(sd_embedding_fwd(ids, p->embed, x_cur, s.M, s.h, s.V, stream));
  for (int l = 0; l < s.L; ++l) {
    LayerActs a = save ? layer_acts(s, base, l, save) : carve(s, base);
    a.x_in = x_cur;
    // read by the GEMM dispatch for every launch of this layer ("model.shared_layers": measurement, sd_hip_debug.h)
    const int lim = save ? g_sd_debug.model_shared_layers_train : g_sd_debug.model_shared_layers;
    SdSharedGpuScope shared(concurrent && (lim < 0 || l < lim) ? 1 : 0);
    char* x_out;
    if (save) x_out = (l + 1 < s.L) ? layer_acts(s, base, l + 1, save).x_in : x_last;
    else x_out = (l + 1 < s.L) ? ((x_cur == pong) ? base : pong) : x_last;
    if (folded)
      RUN(layer_forward_folded(d, s, a, p->layers_host[l], x_out, kv_len, cos_tab, sin_tab, B, T, ssq_a, ssq_b,
                               l + 1 < s.L ? ssq_a : nullptr, stream));
    else
      RUN(layer_forward(d, s, a, p->layers_host[l], x_out, save != SD_SAVE_NONE, kv_len, cos_tab, sin_tab, B, T, stream));
    x_cur = x_out;
  }
  RUN(sd_rmsnorm_fwd(x_last, p->final_norm, xn_f, (float*)rstd_f, s.M, s.h, d->eps, stream));
  if (logits && head_rows) {
    // lm_head only for the rows the loss will read (HF computes all B*T rows, train.py:54-55; the rows whose shifted
    // label is -100 never reach the loss, distillation_loss.py:37-45)
    RUN(sd_embedding_fwd(head_rows, xn_f, xn_rows, n_head_rows, s.h, s.M, stream));
    RUN(sd_gemm_bf16(xn_rows, p->lm_head, logits, nullptr, n_head_rows, s.V, s.h, s.h, s.h, s.V, 0, 0, 0, stream));
  } else if (logits) {
    RUN(sd_gemm_bf16(xn_f, p->lm_head, logits, nullptr, s.M, s.V, s.h, s.h, s.h, s.V, 0, 0, 0, stream));
  }
  return 0;
}

extern "C" int sd_qwen3_backward(const sd_qwen3_dims* d, const sd_qwen3_params* p, const sd_qwen3_params* g,
                                 const int64_t* ids, const int32_t* kv_len, const void* cos_tab, const void* sin_tab,
                                 void* acts, int64_t acts_bytes, void* dlogits, void* scratch, int64_t scratch_bytes, int B,
                                 int T, int accumulate, void* dx0_out, sd_stage_cb on_grads_ready, void* cb_user,
                                 void* side_stream, void* stream) {
  return sd_qwen3_backward_rows(d, p, g, ids, kv_len, cos_tab, sin_tab, acts, acts_bytes, dlogits, nullptr, 0, scratch,
                                scratch_bytes, B, T, accumulate, dx0_out, on_grads_ready, cb_user, side_stream, stream);
}

extern "C" int sd_qwen3_backward_rows(const sd_qwen3_dims* d, const sd_qwen3_params* p, const sd_qwen3_params* g,
                                      const int64_t* ids, const int32_t* kv_len, const void* cos_tab, const void* sin_tab,
                                      void* acts, int64_t acts_bytes, void* dlogits, const int64_t* head_rows,
                                      int n_head_rows, void* scratch, int64_t scratch_bytes, int B, int T, int accumulate,
                                      void* dx0_out, sd_stage_cb on_grads_ready, void* cb_user, void* side_stream,
                                      void* stream) {
  if (d->head_dim != 128) return SD_ERR_UNSUPPORTED;
  if (head_rows && (n_head_rows <= 0 || n_head_rows > B * T)) return SD_ERR_SHAPE;
  if (B <= 0 || T <= 0 || (accumulate & ~(SD_BWD_ACCUMULATE | SD_BWD_RECOMPUTE))) return SD_ERR_SHAPE;
  Sizes s(d, B, T);
  // SD_BWD_RECOMPUTE: `acts` came from a forward with SD_SAVE_LAYER_INPUTS; each layer's forward is run again from its
  // saved input right before its backward (the last layer's buffers are still those of the forward itself)
  const int save = (accumulate & SD_BWD_RECOMPUTE) ? SD_SAVE_LAYER_INPUTS : SD_SAVE_ALL;
  if (acts_bytes < sd_qwen3_acts_bytes(d, B, T, save)) return SD_ERR_WORKSPACE;
  BwdScratch b(s, (char*)scratch);
  if (scratch_bytes < b.total) return SD_ERR_WORKSPACE;
  char* base = (char*)acts;
  const float scale = 0.08838834764831845f;
  char* tail = base + s.body(save);
  char* x_last = tail;
  char* rstd_f = tail + s.x;
  char* xn_f = rstd_f + s.rstd;
  char* xn_rows = xn_f + s.x;
  const int acc = (accumulate & SD_BWD_ACCUMULATE) ? 1 : 0;
#define ACC(ptr) (acc ? (const void*)(ptr) : (const void*)nullptr)
  // A/B switch for measurements ("model.overlap_mask", sd_hip_debug.h): bit0 lm_head dW, bit1 gain reduces, bit2 attention
  // dQ, bit3 grouped per-layer dW, bit4 batched per-layer gain reduce (default all on)
  const int ovl = g_sd_debug.model_overlap_mask;
  hipStream_t s1 = (hipStream_t)stream, s2 = (hipStream_t)side_stream;
  SdEventLease lease;
  if (s2 && !(lease.set = sd_lease_events())) return SD_ERR_WORKSPACE;
  hipEvent_t* g_ev = lease.set ? lease.set->ev : nullptr;  // this call's events
  void* wstream = s2 ? side_stream : stream;  // where weight-gradient GEMMs go
  SdSharedGpuScope shared(s2 ? 1 : 0);  // two streams share the GPU: the persistent dW launches leave CUs to the dX chain
  // main -> side: "this buffer is final"; side -> main: "this layer's dW GEMMs have read their inputs"
#define SIGNAL(i) do { if (s2) { if (hipEventRecord(g_ev[i], s1) != hipSuccess || hipStreamWaitEvent(s2, g_ev[i], 0) != hipSuccess) return SD_ERR_WORKSPACE; } } while (0)
#define JOIN() do { if (s2) { if (hipEventRecord(g_ev[7], s2) != hipSuccess || hipStreamWaitEvent(s1, g_ev[7], 0) != hipSuccess) return SD_ERR_WORKSPACE; } } while (0)

  // lm_head: dxn = dlogits . W ; dW (+)= dlogits^T . xn_f
  const int top = (s.L - 1) & 1;  // buffer set of the last layer (the first one the backward visits)
  SIGNAL(0);  // dlogits (produced on `stream` by the caller) is final: lm_head dW runs beside lm_head dX
  int nsp = 1;
  if (head_rows) {
    // dlogits holds only the n_head_rows rows the forward produced; every other row of d(xn_f) is zero
    RUN(sd_gemm_bf16(dlogits, xn_rows, g->lm_head, ACC(g->lm_head), s.V, s.h, n_head_rows, s.V, s.h, s.h, s.h, 1, 1,
                     (ovl & 1) ? wstream : stream));
    RUN(sd_gemm_bf16_splitk(dlogits, p->lm_head, b.dxb[top], nullptr, n_head_rows, s.h, s.V, s.V, s.h, s.h, 0, 0, 1,
                            b.ws_splitk, b.splitk_bytes, stream));
    RUN(sd_rows_scatter(b.dxb[top], head_rows, b.dxn, n_head_rows, s.M, s.h, stream));
  } else {
    RUN(sd_gemm_bf16(dlogits, xn_f, g->lm_head, ACC(g->lm_head), s.V, s.h, s.M, s.V, s.h, s.h, s.h, 1, 1,
                     (ovl & 1) ? wstream : stream));
    RUN(sd_gemm_bf16_splitk_partial(dlogits, p->lm_head, b.dxn, s.M, s.h, s.V, s.V, s.h, s.h, 0, 1, b.ws_splitk,
                                    b.splitk_bytes, &nsp, stream));
  }
  if (g->embed != g->lm_head && !acc)
    if (hipMemsetAsync(g->embed, 0, (size_t)s.V * s.h * 2, (hipStream_t)stream) != hipSuccess) return SD_ERR_WORKSPACE;
  // the norm backward sums the split-K slabs itself (no separate reduce pass)
#define NORM_BWD(X, W, RSTD, DRES, DX, DW, WS, RS, EV)                                                                   \
  do {                                                                                                                   \
    if (nsp > 1) RUN(sd_rmsnorm_bwd_slabs((const float*)b.ws_splitk, nsp, X, W, RSTD, DRES, DX, DW, acc, WS, s.M, s.h, RS, \
                                          EV, stream));                                                                  \
    else RUN(sd_rmsnorm_bwd2(b.dxn, X, W, RSTD, DRES, DX, DW, acc, WS, s.M, s.h, RS, EV, stream));                         \
  } while (0)
  NORM_BWD(x_last, p->final_norm, (const float*)rstd_f, nullptr, b.dxa[top], g->final_norm, b.ws_norm, nullptr, nullptr);
  JOIN();
  if (on_grads_ready) on_grads_ready(SD_STAGE_HEAD, cb_user);
  // The four weight gradients of a layer run as ONE persistent grouped launch (sd_gemm_grouped_tn: 926 vs 587 TFLOP/s
  // for four separate launches) on the side stream once the layer's dX chain has produced their inputs, i.e. under
  // the chain of the NEXT layer; that layer joins it (event) before it overwrites the gradient buffer they share,
  // and only then is the finished layer reported to the caller.  SD_OVERLAP_MASK bit 3 = 0 keeps four separate GEMMs
  // launched as their inputs appear.
  const bool grouped = (ovl & 8) != 0;
  const bool batch_gains = grouped && (ovl & 16) != 0;  // bit 4: one batched gain-gradient reduce per layer
  int pending = -1;  // layer whose grouped dW is in flight on the side stream
  for (int l = s.L - 1; l >= 0; --l) {
    const int P = l & 1;
    char *dx_in = b.dxa[P], *dx_out = (l == 0 && dx0_out) ? (char*)dx0_out : b.dxa[P ^ 1];
    char *dxb = b.dxb[P], *dqkv = b.dqkv[P], *dgu = b.dgu[P];
    const LayerActs a = layer_acts(s, base, l, save);
    const sd_qwen3_layer& w = p->layers_host[l];
    const sd_qwen3_layer& gw = g->layers_host[l];
    // recompute: this layer's work set was last read by the weight-gradient GEMMs of layer l+2, which `stream` has
    // already waited for (the `pending` join of layer l+1 below)
    if (save == SD_SAVE_LAYER_INPUTS && l != s.L - 1)
      RUN(layer_forward(d, s, a, w, nullptr, true, kv_len, cos_tab, sin_tab, B, T, stream));
    // MLP
    if (!grouped) {
      SIGNAL(0);  // dx_in final
      RUN(sd_gemm_bf16(dx_in, a.act, gw.wdown, ACC(gw.wdown), s.h, s.I, s.M, s.h, s.I, s.I, s.I, 1, 1, wstream));
    }
    // d(act) = dx_in . W_down with the SwiGLU backward in the epilogue: d(act) itself never reaches HBM
    {
      const int rc = sd_gemm_swiglu_bwd(dx_in, w.wdown, a.gu, dgu, s.M, s.I, s.h, stream);
      if (rc == SD_ERR_UNSUPPORTED) {
        RUN(sd_gemm_bf16(dx_in, w.wdown, b.dact, nullptr, s.M, s.I, s.h, s.h, s.I, s.I, 0, 0, 1, stream));
        RUN(sd_swiglu_bwd(b.dact, a.gu, dgu, s.M, s.I, stream));
      } else if (rc) {
        return rc;
      }
    }
    if (!grouped) SIGNAL(1);  // dgu final
    RUN(sd_gemm_bf16_splitk_partial(dgu, w.wgu, b.dxn, s.M, s.h, 2 * s.I, 2 * s.I, s.h, s.h, 0, 1, b.ws_splitk,
                                    b.splitk_bytes, &nsp, stream));
    if (!grouped) RUN(sd_gemm_bf16(dgu, a.xn2, gw.wgu, ACC(gw.wgu), 2 * s.I, s.h, s.M, 2 * s.I, s.h, s.h, s.h, 1, 1, wstream));
    // gain gradients of the layer: the kernels leave per-workgroup partial sums, ONE batched reduce finishes all four
    // (input norm, post-attention norm, q gain, k gain) on the side stream once the layer's last kernel is enqueued
    NORM_BWD(a.x_mid, w.ln2, (const float*)a.rstd2, dx_in, dxb, batch_gains ? nullptr : gw.ln2, b.ws_norm2[P],
             (ovl & 2) ? side_stream : nullptr, s2 ? (void*)g_ev[8] : nullptr);
    if (!grouped) SIGNAL(2);  // dxb final
    // attention
    // d(attention output) = dxb . Wo with delta = rowsum(dO * O) in the epilogue (one 128-column tile = one head)
    const void* o_for_delta = a.ao;
    {
      const int rc = sd_gemm_odx_delta(dxb, w.wo, b.dao, a.ao, s.QD, (float*)b.delta, s.M, T, s.Hq, s.h, stream);
      if (rc == SD_ERR_UNSUPPORTED) RUN(sd_gemm_bf16(dxb, w.wo, b.dao, nullptr, s.M, s.QD, s.h, s.h, s.QD, s.QD, 0, 0, 1, stream));
      else if (rc) return rc;
      else o_for_delta = nullptr;
    }
    if (!grouped) RUN(sd_gemm_bf16(dxb, a.ao, gw.wo, ACC(gw.wo), s.h, s.QD, s.M, s.h, s.QD, s.QD, s.QD, 1, 1, wstream));
    RUN(sd_attn_bwd2(a.qk, a.qk + (int64_t)s.QD * 2, a.qkv + (int64_t)(s.QD + s.KD) * 2, o_for_delta, b.dao, (const float*)a.lse,
                    (float*)b.delta, b.dqk, b.dqk + (int64_t)s.QD * 2, dqkv + (int64_t)(s.QD + s.KD) * 2, kv_len, s.QK,
                    s.QK, s.QKV, s.QD, s.QK, s.QK, s.QKV, B, T, s.Hq, s.Hkv, 128, scale, (ovl & 4) ? side_stream : nullptr, stream));
    RUN(sd_qknorm_rope_bwd2(b.dqk, a.qkv, w.q_gain, w.k_gain, cos_tab, sin_tab, dqkv, batch_gains ? nullptr : gw.q_gain,
                            batch_gains ? nullptr : gw.k_gain, acc, b.ws_qk[P], s.M, T, s.Hq, s.Hkv, d->eps,
                            (ovl & 2) ? side_stream : nullptr, s2 ? (void*)g_ev[9] : nullptr, stream));
    SIGNAL(3);  // dqkv final (and with it dx_in, dgu, dxb of this layer)
    if (grouped) {
      sd_gemm_problem pr[4] = {
          {dqkv, a.xn1, gw.wqkv, s.QKV, s.h, s.h, s.QKV, s.h},    // dW_qkv  [QKV,h]  = dqkv^T . xn1
          {dgu, a.xn2, gw.wgu, 2 * s.I, s.h, s.h, 2 * s.I, s.h},  // dW_gu   [2I,h]   = dgu^T  . xn2
          {dx_in, a.act, gw.wdown, s.h, s.I, s.I, s.h, s.I},      // dW_down [h,I]    = dx_in^T . act
          {dxb, a.ao, gw.wo, s.h, s.QD, s.QD, s.h, s.QD}};        // dW_o    [h,QD]   = dxb^T  . ao
      const int rc = sd_gemm_grouped_tn(pr, 4, s.M, acc, wstream);
      if (rc == SD_ERR_UNSUPPORTED) {
        for (const sd_gemm_problem& q : pr)
          RUN(sd_gemm_bf16(q.A, q.B, q.C, acc ? q.C : nullptr, q.M, q.N, s.M, q.lda, q.ldb, q.ldc, q.ldc, 1, 1, wstream));
      } else if (rc) {
        return rc;
      }
      if (!batch_gains && s2 && hipEventRecord(g_ev[10 + P], s2) != hipSuccess) return SD_ERR_WORKSPACE;
    } else {
      RUN(sd_gemm_bf16(dqkv, a.xn1, gw.wqkv, ACC(gw.wqkv), s.QKV, s.h, s.M, s.QKV, s.h, s.h, s.h, 1, 1, wstream));
    }
    RUN(sd_gemm_bf16_splitk_partial(dqkv, w.wqkv, b.dxn, s.M, s.h, s.QKV, s.QKV, s.h, s.h, 0, 1, b.ws_splitk,
                                    b.splitk_bytes, &nsp, stream));
    if (grouped) {
      // the previous layer's grouped dW reads the buffer this layer is about to overwrite with its output gradient
      if (pending >= 0 && s2 && hipStreamWaitEvent(s1, g_ev[10 + (pending & 1)], 0) != hipSuccess) return SD_ERR_WORKSPACE;
    } else {
      JOIN();  // the layer's dW GEMMs are done before their inputs are overwritten and before the callback
    }
    NORM_BWD(a.x_in, w.ln1, (const float*)a.rstd1, dxb, dx_out, batch_gains ? nullptr : gw.ln1,
             batch_gains ? b.ws_norm1[P] : b.ws_norm, nullptr, nullptr);
    if (batch_gains) {
      // the side stream (after this layer's grouped dW) waits for the partials, reduces, and only then marks the
      // layer finished: the event below now covers the weight AND the gain gradients of layer l
      if (s2 && (hipEventRecord(g_ev[6], s1) != hipSuccess || hipStreamWaitEvent(s2, g_ev[6], 0) != hipSuccess))
        return SD_ERR_WORKSPACE;
      const int nb_n = sd_rmsnorm_bwd_partial_rows(s.M, s.h), nb_q = sd_qknorm_rope_bwd_partial_rows(s.M, s.Hq, s.Hkv);
      const sd_colsum_problem cp[4] = {{(const float*)b.ws_norm2[P], gw.ln2, nb_n, s.h, s.h, acc},
                                       {(const float*)b.ws_norm1[P], gw.ln1, nb_n, s.h, s.h, acc},
                                       {(const float*)b.ws_qk[P], gw.q_gain, nb_q, 128, 256, acc},
                                       {(const float*)b.ws_qk[P] + 128, gw.k_gain, nb_q, 128, 256, acc}};
      RUN(sd_colsum_reduce_batch(cp, 4, wstream));
      if (s2 && hipEventRecord(g_ev[10 + P], s2) != hipSuccess) return SD_ERR_WORKSPACE;
    }
    if (grouped) {
      if (pending >= 0 && on_grads_ready) on_grads_ready(pending, cb_user);
      pending = l;
    } else if (on_grads_ready) {
      on_grads_ready(l, cb_user);
    }
  }
  if (grouped && pending >= 0) {
    if (s2 && hipStreamWaitEvent(s1, g_ev[10 + (pending & 1)], 0) != hipSuccess) return SD_ERR_WORKSPACE;
    if (on_grads_ready) on_grads_ready(pending, cb_user);
  }
  JOIN();  // everything the side stream was given (gain reduces, dQ) before the call returns
  if (!dx0_out) RUN(sd_embedding_bwd(ids, b.dxa[1], g->embed, s.M, s.h, s.V, 1.0f, stream));
  if (on_grads_ready) on_grads_ready(SD_STAGE_EMBED, cb_user);
#undef ACC
#undef NORM_BWD
#undef SIGNAL
#undef JOIN
  return 0;
}
