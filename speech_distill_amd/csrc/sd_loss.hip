// Fused temperature-scaled KL + CE distillation loss over the expanded speech-token vocabulary,
// forward and backward, for gfx950.  HBM-bound: one 1024-thread workgroup streams one logits row.
//
// Replaces DistillationLoss.forward and its autograd backward:
//   /root/reference/distillation_loss.py:31-45   causal shift + valid-row predicate (no compaction
//                                                copy here: the predicate is evaluated per row)
//   /root/reference/distillation_loss.py:47-53   N == 0 -> zeros
//   /root/reference/distillation_loss.py:56-71   dense KL (batchmean) * T^2, teacher CE monitor
//   /root/reference/distillation_loss.py:73-118  sparse top-K KL, approximate teacher monitor
//   /root/reference/distillation_loss.py:123-128 CE task loss, alpha mix
// Gradient:
//   d total / d s = a (softmax(s) - e_y)/N + (1-a) T (softmax(s/T) - q)/N   on valid rows, else 0.
//
// Forward keeps per-row normalisers (lse at T=1, lse at T, teacher lse at T) so that backward is one
// read of the logits and one write of the gradient.  All reductions are deterministic (per-row
// partials + a fixed-order final reduce); no float atomics.
#include "sd_common.cuh"
#include "../../include/sd_hip.h"
#include "sd_prof.h"

namespace {

constexpr int NT = 1024;
constexpr int KMAX = 1024;

template <typename T> struct Ld8;
template <> struct Ld8<bf16> {
  typedef bf16x8 Raw;  // 8 elements as loaded (prefetched chunks wait in registers in this form)
  static SD_DEV Raw raw(const bf16* p) { return *(const bf16x8*)p; }
  static SD_DEV void cvt(const Raw& v, float* f) {
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
  }
  static SD_DEV void load(const bf16* p, float* f) {
    bf16x8 v = *(const bf16x8*)p;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
  }
  static SD_DEV void store(bf16* p, const float* f) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)f[e];
    *(bf16x8*)p = v;
  }
};
template <> struct Ld8<float> {
  struct Raw { f32x4 a, b; };
  static SD_DEV Raw raw(const float* p) { return Raw{*(const f32x4*)p, *(const f32x4*)(p + 4)}; }
  static SD_DEV void cvt(const Raw& v, float* f) {
    f[0] = v.a[0]; f[1] = v.a[1]; f[2] = v.a[2]; f[3] = v.a[3]; f[4] = v.b[0]; f[5] = v.b[1]; f[6] = v.b[2]; f[7] = v.b[3];
  }
  static SD_DEV void load(const float* p, float* f) {
    f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
    f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
  }
  static SD_DEV void store(float* p, const float* f) {
    *(f32x4*)p = f32x4{f[0], f[1], f[2], f[3]};
    *(f32x4*)(p + 4) = f32x4{f[4], f[5], f[6], f[7]};
  }
};

// row -> (b, t); label / mask are taken at t+1 (the causal shift).  T == 0: the caller already selected and
// shifted the rows (sd_kdloss_fwd_rows): label / mask are taken at `row` itself.
SD_DEV bool row_valid(const int64_t* labels, const uint8_t* mask, int row, int T, long* y) {
  int at = row;
  if (T) {
    const int t = row % T;
    if (t == T - 1) return false;
    at = row + 1;
  }
  const long lab = labels[at];
  *y = lab;
  if (lab == -100) return false;
  if (mask && !mask[at]) return false;
  return true;
}

struct RowStats {  // 8 floats per row
  float lse1, lseT, t_lseT, valid, task, distill, teacher, hits;
};

// merge (m, s) online-softmax partials across the block; returns the block totals to all threads
template <int NF>
SD_DEV void block_merge(float& m, float& s1, float& sT, float invT, float* sc) {
  const float M = block_max<NF>(m, sc);
  const float f1 = (m == -INFINITY) ? 0.f : __expf(m - M);
  const float fT = (m == -INFINITY) ? 0.f : __expf((m - M) * invT);
  s1 = block_sum<NF>(s1 * f1, sc);
  sT = block_sum<NF>(sT * fT, sc);
  m = M;
}

// T2: temperature == 2 (the reference's default, train.py:488-491).  Then exp(z) = exp(z / 2)^2, so the sum-exp at T = 1
// comes from the T = 2 exponential by one multiplication: half the transcendentals of a kernel that is bound by them
// (129 us = 3.8 TB/s before, with two v_exp_f32 per logit at a quarter of the vector rate).
// NF threads per row: 512 (four workgroups per CU, 128 chunks of a CU in flight) whenever the K teacher entries fit one
// thread each, else 1024.  DENSE: teacher logits instead of top-K (its prefetch registers only exist in that variant).
template <typename T, bool T2, int NF, bool DENSE>
__global__ __launch_bounds__(NF) void kd_fwd_kernel(const T* __restrict__ S, const T* __restrict__ Tl,
                                                    const _Float16* __restrict__ topv, const int32_t* __restrict__ topi,
                                                    const int64_t* __restrict__ labels, const uint8_t* __restrict__ mask,
                                                    RowStats* __restrict__ stats, int rows, int Tlen, int V, int K,
                                                    float temperature) {
  __shared__ float sc[32];
  const int row = blockIdx.x;
  long y = 0;
  const bool valid = row_valid(labels, mask, row, Tlen, &y);
  if (!valid || y < 0 || y >= V) {
    if (threadIdx.x == 0) stats[row] = RowStats{0, 0, 0, 0, 0, 0, 0, 0};
    return;
  }
  const float invT = 1.f / temperature;
  const float k1 = 1.4426950408889634f, kT = invT * k1;  // log2(e), log2(e) / T
  const T* s = S + (long)row * V;
  // ---- student normalisers (online max / sum-exp at T=1 and at T)
  float m = -INFINITY, s1 = 0.f, sT = 0.f;
  // ---- dense teacher: max, sum-exp at 1 and T, cross = sum exp((t-mt)/T) (t - s)
  float mt = -INFINITY, t1 = 0.f, tT = 0.f, cross = 0.f;
  const T* tl = DENSE ? Tl + (long)row * V : nullptr;
  // The row is streamed with PF chunks per thread in flight: with one (load, wait, compute) per trip the 32 waves of a CU
  // keep 32 KB outstanding, which at ~2 us of loaded HBM latency is 4 TB/s whatever the kernel does (measured 3.9-4.1).
  // Each thread still sees its chunks in the same order: the sums are bit for bit the ones of the plain loop.
  constexpr int PF = 4;
  typedef typename Ld8<T>::Raw Raw;
  const int stride = NF * 8;
  Raw sbuf[PF], tbuf[DENSE ? PF : 1];
  // every load is UNCONDITIONAL (a chunk past the end re-reads the row's last one and is never used): with loads behind
  // branches hipcc's wait insertion falls back to vmcnt(0) at every use and drains the chunks in flight
  const int clast = ((V - 1) >> 3) << 3;  // start of the row's last chunk
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    const int cj = min((int)threadIdx.x * 8 + j * stride, clast);
    sbuf[j] = Ld8<T>::raw(s + cj);
    if constexpr (DENSE) tbuf[j] = Ld8<T>::raw(tl + cj);
  }
  for (int c0 = threadIdx.x * 8; c0 < V; c0 += PF * stride) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int c = c0 + j * stride;
      float f[8], g[8];
      Ld8<T>::cvt(sbuf[j], f);
      if constexpr (DENSE) Ld8<T>::cvt(tbuf[j], g);
      const int cn = min(c + PF * stride, clast);  // refill the slot before the arithmetic
      sbuf[j] = Ld8<T>::raw(s + cn);
      if constexpr (DENSE) tbuf[j] = Ld8<T>::raw(tl + cn);
      if (c >= V) continue;
      float cm = f[0];
#pragma unroll
      for (int e = 1; e < 8; ++e) cm = fmaxf(cm, f[e]);
      if (cm > m) {
        const float rT = __expf((m - cm) * invT), r1 = T2 ? rT * rT : __expf(m - cm);
        s1 *= r1; sT *= rT; m = cm;
      }
      // exp((f - m) / T) = exp2(f * kT - m * kT): one fused multiply-add per exponential argument
      const float mkT = -m * kT, mk1 = -m * k1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float q = __builtin_amdgcn_exp2f(__builtin_fmaf(f[e], kT, mkT));
        sT += q;
        s1 += T2 ? q * q : __builtin_amdgcn_exp2f(__builtin_fmaf(f[e], k1, mk1));
      }
      if constexpr (DENSE) {
        float tm = g[0];
#pragma unroll
        for (int e = 1; e < 8; ++e) tm = fmaxf(tm, g[e]);
        if (tm > mt) {
          const float rT = __expf((mt - tm) * invT), r1 = T2 ? rT * rT : __expf(mt - tm);
          t1 *= r1; tT *= rT; cross *= rT; mt = tm;
        }
        const float tkT = -mt * kT, tk1 = -mt * k1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float w = __builtin_amdgcn_exp2f(__builtin_fmaf(g[e], kT, tkT));
          t1 += T2 ? w * w : __builtin_amdgcn_exp2f(__builtin_fmaf(g[e], k1, tk1));
          tT += w;
          cross += w * (g[e] - f[e]);
        }
      }
    }
  }
  block_merge<NF>(m, s1, sT, invT, sc);
  const float lse1 = m + __logf(s1);
  const float lseT = m * invT + __logf(sT);
  const float sy = (float)s[y];
  const float task = lse1 - sy;
  float distill = 0.f, teacher = 0.f, hits = 0.f, t_lseT = 0.f;
  if constexpr (DENSE) {
    const float MT = block_max<NF>(mt, sc);
    const float f1 = (mt == -INFINITY) ? 0.f : __expf(mt - MT);
    const float fT = (mt == -INFINITY) ? 0.f : __expf((mt - MT) * invT);
    t1 = block_sum<NF>(t1 * f1, sc);
    tT = block_sum<NF>(tT * fT, sc);
    cross = block_sum<NF>(cross * fT, sc);
    t_lseT = MT * invT + __logf(tT);
    // sum_v q (log q - log p) = (1/T) sum_v q (t - s) - t_lseT + lseT      (sum_v q = 1)
    distill = cross / tT * invT - t_lseT + lseT;
    teacher = MT + __logf(t1) - (float)tl[y];
    hits = 1.f;
  } else {
    // sparse: q = softmax(v/T) over the K stored teacher log-probs (renormalised over top-K)
    const long kb = (long)row * K;
    const bool act = threadIdx.x < K;
    const float v = act ? (float)topv[kb + threadIdx.x] : -INFINITY;
    const int idx = act ? topi[kb + threadIdx.x] : 0;
    const float mv = block_max<NF>(v, sc);
    const float ev = act ? __expf((v - mv) * invT) : 0.f;
    const float sv = block_sum<NF>(ev, sc);
    const float logq = (v - mv) * invT - __logf(sv);
    float term = 0.f, hv = 0.f, hc = 0.f;
    if (act) {
      const int ii = idx < 0 ? 0 : (idx >= V ? V - 1 : idx);
      const float logp = (float)s[ii] * invT - lseT;
      term = (ev / sv) * (logq - logp);
      if ((long)idx == y) { hv = v; hc = 1.f; }
    }
    distill = block_sum<NF>(term, sc);
    teacher = block_sum<NF>(hv, sc);   // sum of teacher log-probs at the label, over hits
    hits = block_sum<NF>(hc, sc);
  }
  if (threadIdx.x == 0) stats[row] = RowStats{lse1, lseT, t_lseT, 1.f, task, distill, teacher, hits};
}

// out[0..5] = total, task, distill, teacher, N, n_hits      (single block, fixed order)
__global__ __launch_bounds__(NT) void kd_finalize_kernel(const RowStats* __restrict__ stats, float* __restrict__ out,
                                                         int rows, float temperature, float alpha, int dense) {
  __shared__ float sc[32];
  float n = 0, task = 0, dist = 0, teach = 0, hits = 0;
  for (int r = threadIdx.x; r < rows; r += NT) {
    const RowStats st = stats[r];
    n += st.valid; task += st.task; dist += st.distill; teach += st.teacher; hits += st.hits;
  }
  n = block_sum<NT>(n, sc); task = block_sum<NT>(task, sc); dist = block_sum<NT>(dist, sc);
  teach = block_sum<NT>(teach, sc); hits = block_sum<NT>(hits, sc);
  if (threadIdx.x == 0) {
    if (n == 0.f) {
      out[0] = out[1] = out[2] = out[3] = 0.f; out[4] = 0.f; out[5] = 0.f;
    } else {
      const float tk = task / n;
      const float ds = dist / n * temperature * temperature;
      const float tc = dense ? teach / n : (hits > 0.f ? -teach / hits : 0.f);
      out[0] = alpha * tk + (1.f - alpha) * ds; out[1] = tk; out[2] = ds; out[3] = tc; out[4] = n; out[5] = hits;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void kd_bwd_kernel(const T* S, const T* __restrict__ Tl,
                                                    const _Float16* __restrict__ topv, const int32_t* __restrict__ topi,
                                                    const int64_t* __restrict__ labels, const RowStats* __restrict__ stats,
                                                    const float* __restrict__ loss_out, const float* __restrict__ grad_out,
                                                    T* G, int rows, int Tlen, int V, int K, float temperature,
                                                    float alpha) {
  __shared__ float sc[32];
  __shared__ float k_q[KMAX];
  __shared__ float k_s[KMAX];
  __shared__ int k_i[KMAX];
  const int row = blockIdx.x;
  const RowStats st = stats[row];
  T* g = G + (long)row * V;
  if (st.valid == 0.f) {
    float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = threadIdx.x * 8; c < V; c += NT * 8) Ld8<T>::store(g + c, z);
    return;
  }
  const float invT = 1.f / temperature;
  const float N = loss_out[4];
  const float go = grad_out ? grad_out[0] : 1.f;
  const float a1 = go * alpha / N;                       // * softmax(s)
  const float aT = go * (1.f - alpha) * temperature / N;  // * softmax(s/T)  and  * q
  const long y = labels[row + (Tlen ? 1 : 0)];
  const T* s = S + (long)row * V;
  const T* tl = Tl ? Tl + (long)row * V : nullptr;
  const bool sparse = (tl == nullptr);
  if (sparse) {
    // gather everything the fix-up needs BEFORE the dense pass (G may alias S)
    const long kb = (long)row * K;
    const bool act = threadIdx.x < K;
    const float v = act ? (float)topv[kb + threadIdx.x] : -INFINITY;
    const int idx = act ? topi[kb + threadIdx.x] : -1;
    const float mv = block_max<NT>(v, sc);
    const float ev = act ? __expf((v - mv) * invT) : 0.f;
    const float sv = block_sum<NT>(ev, sc);
    if (act) {
      k_q[threadIdx.x] = ev / sv;
      k_i[threadIdx.x] = idx;
      k_s[threadIdx.x] = (idx >= 0 && idx < V) ? (float)s[idx] : 0.f;
    }
    __syncthreads();
  }
  for (int c = threadIdx.x * 8; c < V; c += NT * 8) {
    float f[8], o[8];
    Ld8<T>::load(s + c, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = a1 * __expf(f[e] - st.lse1) + aT * __expf(f[e] * invT - st.lseT);
    if (tl) {
      float q[8];
      Ld8<T>::load(tl + c, q);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] -= aT * __expf(q[e] * invT - st.t_lseT);
    }
    if (y >= c && y < c + 8) o[y - c] -= a1;
    Ld8<T>::store(g + c, o);
  }
  if (sparse) {
    __syncthreads();  // every dense store of this row is complete (vmcnt(0) + barrier) before the fix-up
    if (threadIdx.x < K) {
      const int idx = k_i[threadIdx.x];
      bool first = idx >= 0 && idx < V;
      float qs = 0.f;
      for (int k2 = 0; k2 < K; ++k2) {
        if (k_i[k2] == idx) {
          if (k2 < (int)threadIdx.x) first = false;
          qs += k_q[k2];  // duplicates of an index sum their q (gather semantics of the reference)
        }
      }
      if (first) {
        const float sv = k_s[threadIdx.x];
        float o = a1 * __expf(sv - st.lse1) + aT * __expf(sv * invT - st.lseT) - aT * qs;
        if ((long)idx == y) o -= a1;
        g[idx] = (T)o;
      }
    }
  }
}

template <typename T>
int run_fwd(const void* S, const void* Tl, const void* topv, const void* topi, const int64_t* labels, const uint8_t* mask,
            void* stats, float* out, int B, int Tlen, int V, int K, float temperature, float alpha, hipStream_t st) {
  const int rows = Tlen ? B * Tlen : B;  // Tlen == 0: B pre-selected rows
  SdProfScope prof(SD_K_LOSS_FWD, (double)rows * V * sizeof(T) * (Tl ? 2 : 1), st);
  // the symbol as rocprofv3 --kernel-trace prints it: kd_fwd_kernel<T, T2, NF, DENSE> (same dispatch as below)
  SD_PROF_LABEL("kd_fwd_kernel<%s, %s, %d, %s>", sizeof(T) == 2 ? "__bf16" : "float", temperature == 2.0f ? "true" : "false",
                (Tl || K <= 512) ? 512 : 1024, Tl ? "true" : "false");
#define SD_KD_FWD(T2_, NF_, DENSE_)                                                                                      \
  hipLaunchKernelGGL((kd_fwd_kernel<T, T2_, NF_, DENSE_>), dim3(rows), dim3(NF_), 0, st, (const T*)S, (const T*)Tl,        \
                     (const _Float16*)topv, (const int32_t*)topi, labels, mask, (RowStats*)stats, rows, Tlen, V, K,          \
                     temperature)
  const bool t2 = temperature == 2.0f;
  if (Tl) { if (t2) SD_KD_FWD(true, 512, true); else SD_KD_FWD(false, 512, true); }
  else if (K <= 512) { if (t2) SD_KD_FWD(true, 512, false); else SD_KD_FWD(false, 512, false); }
  else { if (t2) SD_KD_FWD(true, 1024, false); else SD_KD_FWD(false, 1024, false); }
#undef SD_KD_FWD
  SD_CHECK_LAUNCH();
  hipLaunchKernelGGL(kd_finalize_kernel, dim3(1), dim3(NT), 0, st, (const RowStats*)stats, out, rows, temperature, alpha,
                     Tl ? 1 : 0);
  SD_CHECK_LAUNCH();
  return 0;
}

template <typename T>
int run_bwd(const void* S, const void* Tl, const void* topv, const void* topi, const int64_t* labels, const void* stats,
            const float* out, const float* go, void* G, int B, int Tlen, int V, int K, float temperature, float alpha,
            hipStream_t st) {
  const int rows = Tlen ? B * Tlen : B;
  SdProfScope prof(SD_K_LOSS_BWD, (double)rows * V * sizeof(T) * (Tl ? 3 : 2), st);
  SD_PROF_LABEL("kd_bwd_kernel<%s>", sizeof(T) == 2 ? "__bf16" : "float");
  hipLaunchKernelGGL((kd_bwd_kernel<T>), dim3(rows), dim3(NT), 0, st, (const T*)S, (const T*)Tl, (const _Float16*)topv,
                     (const int32_t*)topi, labels, (const RowStats*)stats, out, go, (T*)G, rows, Tlen, V, K, temperature,
                     alpha);
  SD_CHECK_LAUNCH();
  return 0;
}

int check(const void* S, const void* Tl, const void* topv, const void* topi, int B, int Tlen, int V, int K, int dtype) {
  if (B <= 0 || Tlen < 0 || V <= 0) return SD_ERR_SHAPE;
  if (V & 7) return SD_ERR_ALIGN;
  if ((uintptr_t)S & 15) return SD_ERR_ALIGN;
  if (dtype != SD_DTYPE_BF16 && dtype != SD_DTYPE_F32) return SD_ERR_UNSUPPORTED;
  if (!Tl && !(topv && topi)) return SD_ERR_NO_TEACHER;
  if (!Tl && (K <= 0 || K > KMAX)) return SD_ERR_SHAPE;
  return 0;
}

}  // namespace

extern "C" int64_t sd_kdloss_stats_bytes(int B, int T) { return (int64_t)B * T * sizeof(RowStats); }

extern "C" int sd_kdloss_fwd_rows(const void* student_logits, const void* teacher_logits, const void* top_k_v,
                                  const void* top_k_i, const int64_t* row_labels, void* row_stats, float* loss_out, int R,
                                  int V, int K, float temperature, float alpha, int dtype, void* stream) {
  if (int e = check(student_logits, teacher_logits, top_k_v, top_k_i, R, 0, V, K, dtype)) return e;
  if (dtype == SD_DTYPE_BF16)
    return run_fwd<bf16>(student_logits, teacher_logits, top_k_v, top_k_i, row_labels, nullptr, row_stats, loss_out, R, 0,
                         V, K, temperature, alpha, (hipStream_t)stream);
  return run_fwd<float>(student_logits, teacher_logits, top_k_v, top_k_i, row_labels, nullptr, row_stats, loss_out, R, 0,
                        V, K, temperature, alpha, (hipStream_t)stream);
}

extern "C" int sd_kdloss_bwd_rows(const void* student_logits, const void* teacher_logits, const void* top_k_v,
                                  const void* top_k_i, const int64_t* row_labels, const void* row_stats,
                                  const float* loss_out, const float* grad_total, void* grad_logits, int R, int V, int K,
                                  float temperature, float alpha, int dtype, void* stream) {
  if (int e = check(student_logits, teacher_logits, top_k_v, top_k_i, R, 0, V, K, dtype)) return e;
  if ((uintptr_t)grad_logits & 15) return SD_ERR_ALIGN;
  if (dtype == SD_DTYPE_BF16)
    return run_bwd<bf16>(student_logits, teacher_logits, top_k_v, top_k_i, row_labels, row_stats, loss_out, grad_total,
                         grad_logits, R, 0, V, K, temperature, alpha, (hipStream_t)stream);
  return run_bwd<float>(student_logits, teacher_logits, top_k_v, top_k_i, row_labels, row_stats, loss_out, grad_total,
                        grad_logits, R, 0, V, K, temperature, alpha, (hipStream_t)stream);
}

extern "C" int sd_kdloss_fwd(const void* student_logits, const void* teacher_logits, const void* top_k_v,
                             const void* top_k_i, const int64_t* labels, const uint8_t* speech_mask, void* row_stats,
                             float* loss_out, int B, int T, int V, int K, float temperature, float alpha, int dtype,
                             void* stream) {
  if (int e = check(student_logits, teacher_logits, top_k_v, top_k_i, B, T, V, K, dtype)) return e;
  if (dtype == SD_DTYPE_BF16)
    return run_fwd<bf16>(student_logits, teacher_logits, top_k_v, top_k_i, labels, speech_mask, row_stats, loss_out, B, T,
                         V, K, temperature, alpha, (hipStream_t)stream);
  return run_fwd<float>(student_logits, teacher_logits, top_k_v, top_k_i, labels, speech_mask, row_stats, loss_out, B, T,
                        V, K, temperature, alpha, (hipStream_t)stream);
}

extern "C" int sd_kdloss_bwd(const void* student_logits, const void* teacher_logits, const void* top_k_v,
                             const void* top_k_i, const int64_t* labels, const void* row_stats, const float* loss_out,
                             const float* grad_total, void* grad_logits, int B, int T, int V, int K, float temperature,
                             float alpha, int dtype, void* stream) {
  if (int e = check(student_logits, teacher_logits, top_k_v, top_k_i, B, T, V, K, dtype)) return e;
  if ((uintptr_t)grad_logits & 15) return SD_ERR_ALIGN;
  if (dtype == SD_DTYPE_BF16)
    return run_bwd<bf16>(student_logits, teacher_logits, top_k_v, top_k_i, labels, row_stats, loss_out, grad_total,
                         grad_logits, B, T, V, K, temperature, alpha, (hipStream_t)stream);
  return run_bwd<float>(student_logits, teacher_logits, top_k_v, top_k_i, labels, row_stats, loss_out, grad_total,
                        grad_logits, B, T, V, K, temperature, alpha, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------- loss rows
// distillation_loss.py:31-45: the loss reads position t of sequence b iff t < T-1, labels[b][t+1] != -100 and (with a
// speech mask) speech_mask[b][t+1] != 0.  One workgroup compacts the flat indices b*T+t of those rows IN ORDER (ballot
// prefix inside a wave, wave totals through LDS), writes the label each one predicts, and checks that the attention
// masks are valid-prefix (right-padded) masks -- everything the host needs before it can size the lm_head GEMMs, in one
// launch and one 8-byte read instead of ~15 tiny torch kernels.
namespace {

__global__ __launch_bounds__(1024) void loss_rows_kernel(const long long* __restrict__ labels,
                                                         const long long* __restrict__ speech_mask,
                                                         const long long* __restrict__ mask_a,
                                                         const long long* __restrict__ mask_b, long long* __restrict__ rows,
                                                         long long* __restrict__ row_labels, int* __restrict__ meta, int B,
                                                         int T) {
  __shared__ int wave_tot[16];
  __shared__ int base_s, bad_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) { base_s = 0; bad_s = 0; }
  __syncthreads();
  const long n = (long)B * T;
  int bad = 0;
  for (long c0 = 0; c0 < n; c0 += 1024) {
    const long e = c0 + tid;
    bool valid = false;
    long long lab = -100;
    if (e < n) {
      const int t = (int)(e % T);
      if (t + 1 < T) {
        lab = labels[e + 1];
        valid = lab != -100 && (!speech_mask || speech_mask[e + 1] != 0);
      }
      if (t > 0) {
        if (mask_a && mask_a[e] != 0 && mask_a[e - 1] == 0) bad = 1;
        if (mask_b && mask_b[e] != 0 && mask_b[e - 1] == 0) bad = 1;
      }
    }
    const unsigned long long bal = __ballot(valid);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wv] = __popcll(bal);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wv; ++w) off += wave_tot[w];
    if (valid) {
      rows[off + before] = e;
      row_labels[off + before] = lab;
    }
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += wave_tot[w];
      base_s += tot;
    }
    __syncthreads();
  }
  if (bad) atomicOr(&bad_s, 1);
  __syncthreads();
  if (tid == 0) {
    meta[0] = base_s;
    meta[1] = bad_s;
  }
}

}  // namespace

extern "C" int sd_loss_rows(const int64_t* labels, const int64_t* speech_mask, const int64_t* mask_a, const int64_t* mask_b,
                            int64_t* rows, int64_t* row_labels, int32_t* meta, int B, int T, void* stream) {
  if (B <= 0 || T <= 0 || (long)B * T > (1l << 30)) return SD_ERR_SHAPE;
  if (!labels || !rows || !row_labels || !meta) return SD_ERR_SHAPE;
  SD_PROF_LABEL("loss_rows B=%d T=%d", B, T);
  SdProfScope prof(SD_K_MISC, 8.0 * B * T, (hipStream_t)stream);
  hipLaunchKernelGGL(loss_rows_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const long long*)labels,
                     (const long long*)speech_mask, (const long long*)mask_a, (const long long*)mask_b, (long long*)rows,
                     (long long*)row_labels, meta, B, T);
  SD_CHECK_LAUNCH();
  return 0;
}
