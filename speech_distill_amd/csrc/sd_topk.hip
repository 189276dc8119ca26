// Teacher log-softmax + exact top-K select over the vocabulary, one pass family per row, gfx950.
//
// Replaces /root/reference/train.py:80-91 (on-the-fly sparse extraction inside compute_loss) and
// /root/reference/extract_teacher_logits.py:114-129 (offline extraction):
//     logits[..., :V] -> log_softmax (T=1) -> topk(K), sorted descending -> values fp16, indices int32
// Selection is an exact MSB-first radix select on order-preserving integer keys of the raw logits
// (log-softmax is monotone, so selecting on logits == selecting on log-probs); ties at the K-th
// value go to the LOWEST indices.  One 512-thread workgroup per row, two resident per CU; the row is re-read from
// L2/Infinity Cache by the later radix passes, so HBM sees it about once.
#include <stdlib.h>
#include "sd_common.cuh"
#include "../../include/sd_hip.h"
#include "sd_prof.h"
#include "sd_debug.h"

namespace {

constexpr int KMAX = 1024;

SD_DEV uint32_t f2key(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
SD_DEV float key2f(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
  return __uint_as_float(u);
}

template <typename T> struct Raw8;
template <> struct Raw8<bf16> { typedef bf16x8 R; };
template <> struct Raw8<float> { struct R { f32x4 a, b; }; };
SD_DEV bf16x8 raw8(const bf16* p) { return *(const bf16x8*)p; }
SD_DEV Raw8<float>::R raw8(const float* p) { return Raw8<float>::R{*(const f32x4*)p, *(const f32x4*)(p + 4)}; }
SD_DEV void cvt8(const bf16x8& v, float* f) {
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
}
SD_DEV void cvt8(const Raw8<float>::R& v, float* f) {
  f[0] = v.a[0]; f[1] = v.a[1]; f[2] = v.a[2]; f[3] = v.a[3]; f[4] = v.b[0]; f[5] = v.b[1]; f[6] = v.b[2]; f[7] = v.b[3];
}
template <typename T> SD_DEV void load8(const T* p, float* f);
template <> SD_DEV void load8<bf16>(const bf16* p, float* f) {
  bf16x8 v = *(const bf16x8*)p;
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
}
template <> SD_DEV void load8<float>(const float* p, float* f) {
  f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
  f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
}

// wave-aggregated histogram increment: lanes with the same bin share one LDS atomic
SD_DEV void hist_add(int* hist, int bin, bool active) {
  unsigned long long todo = __ballot(active);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int b = __shfl(bin, leader, 64);
    const unsigned long long same = __ballot(active && bin == b) & todo;
    if (lane_id() == leader) atomicAdd(&hist[b], __popcll(same));
    todo &= ~same;
  }
}

// inclusive-from-the-top scan of a 256-bin histogram by wave 0: finds the bin holding the kth-largest.
// Returns (bin, remaining kth inside that bin) through shared scalars.
SD_DEV void pick_bin(const int* hist, int kth, int* sel_bin, int* sel_kth) {
  if (threadIdx.x < 64) {
    const int l = threadIdx.x;
    // lane l owns bins 255-4l .. 252-4l (descending order)
    int c[4], tot = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { c[j] = hist[255 - (4 * l + j)]; tot += c[j]; }
    int incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (l >= o) incl += v;
    }
    int above = incl - tot;  // elements in strictly higher bins than this lane's first bin
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (above < kth && kth <= above + c[j]) { *sel_bin = 255 - (4 * l + j); *sel_kth = kth - above; }
      above += c[j];
    }
  }
}

constexpr int CAP = 2048;  // candidate capacity of the fast path

// descending bitonic sort of items[0..P2) (P2 a power of two), all NT threads participate
template <int NT>
SD_DEV void bitonic_desc(unsigned long long* items, int P2) {
  for (int size = 2; size <= P2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = threadIdx.x; i < P2 / 2; i += NT) {
        const int lo = ((i / stride) * stride * 2) + (i % stride);
        const int hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long a = items[lo], b = items[hi];
        if (desc ? (a < b) : (a > b)) { items[lo] = b; items[hi] = a; }
      }
      __syncthreads();
    }
  }
}

// Fast path.  Thread t's maximum over its own strided share of the row is one of the row's values, so
// the K-th largest of the NT thread maxima (t0) is a lower bound of the row's K-th largest value:
// every top-K element has key >= t0, and for non-degenerate rows only ~K..2K elements do.  Those few
// candidates are sorted exactly (value desc, index asc).  Rows with more than CAP candidates (massive
// ties) take the general MSB-first radix select below.
// NT threads per row: several workgroups are resident per CU (NT = 512: 4), so one row's select / sort phases (all
// barriers, no memory traffic) run beside other rows' streaming reads (1024 threads, one workgroup per CU: 259 us at
// R = 1536, V = 159 488; 512 or 256 threads: 195 us).  Measured and dropped: keeping every thread's 4 largest items in
// registers during the first read so that the row is read once (the insertion test fires in some lane of every wave
// for every chunk at 39 chunks per thread: 247 us).
template <typename T, int NPASS, int NT>
__global__ __launch_bounds__(NT) void topk_kernel(const T* __restrict__ X, _Float16* __restrict__ outv,
                                                  int32_t* __restrict__ outi, float* __restrict__ lse_out, int rows,
                                                  long row_stride, int V, int K, int KP) {
  __shared__ float sc[32];
  __shared__ int hist[256];
  __shared__ int sel_bin, sel_kth, n_out, tie_base;
  __shared__ int wave_cnt[NT / 64];
  __shared__ uint32_t tkeys[NT];
  __shared__ unsigned long long items[CAP];
  const int row = blockIdx.x;
  const T* x = X + (long)row * row_stride;
  const int lane = lane_id(), wv = threadIdx.x >> 6;

  // ---- pass A: log-sum-exp statistics and per-thread maxima (one HBM read of the row)
  if (threadIdx.x == 0) { n_out = 0; tie_base = 0; }
  float m = -INFINITY, s = 0.f;
  // Rows of up to MAXCH * NT * 8 elements (the 159 488-entry vocabulary at NT = 512): the maximum of every 8-element
  // chunk stays in a register, so that pass B re-reads only the chunks that can hold a candidate (~K of 19 936) instead
  // of the whole row.  Longer rows take the plain loops.
  constexpr int MAXCH = (NT == 512) ? 40 : (NT == 1024 ? 20 : 0);
  const bool keep_max = MAXCH > 0 && V <= MAXCH * NT * 8;
  float cmx[MAXCH > 0 ? MAXCH : 1];
  if (keep_max) {
    // PF chunks per thread in flight (one load per trip leaves a CU's 32 waves with 32 KB outstanding: ~4 TB/s at the
    // loaded HBM latency, whatever the arithmetic); the chunks are consumed in the same order
    constexpr int PF = 4;
    typename Raw8<T>::R buf[PF];
    // every load is UNCONDITIONAL (a chunk past the end re-reads the row's last one and is not used): with loads behind
    // branches hipcc's wait insertion falls back to vmcnt(0) at every use and drains the chunks in flight
    const int clast = ((V - 1) >> 3) << 3;  // start of the row's last chunk
#pragma unroll
    for (int j = 0; j < PF; ++j) buf[j] = raw8(x + min((j * NT + (int)threadIdx.x) * 8, clast));
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int c = (i * NT + (int)threadIdx.x) * 8;
      cmx[i] = -INFINITY;
      float f[8];
      cvt8(buf[i % PF], f);
      if (i + PF < MAXCH) buf[i % PF] = raw8(x + min(((i + PF) * NT + (int)threadIdx.x) * 8, clast));
      if (c < V) {
        float cm = f[0];
#pragma unroll
        for (int e = 1; e < 8; ++e) cm = fmaxf(cm, f[e]);
        if (cm > m) { s *= __expf(m - cm); m = cm; }
#pragma unroll
        for (int e = 0; e < 8; ++e) s += __expf(f[e] - m);
        cmx[i] = cm;
      }
    }
  } else {
    for (int c = threadIdx.x * 8; c < V; c += NT * 8) {
      float f[8];
      load8<T>(x + c, f);
      float cm = f[0];
#pragma unroll
      for (int e = 1; e < 8; ++e) cm = fmaxf(cm, f[e]);
      if (cm > m) { s *= __expf(m - cm); m = cm; }
#pragma unroll
      for (int e = 0; e < 8; ++e) s += __expf(f[e] - m);
    }
  }
  tkeys[threadIdx.x] = (m == -INFINITY) ? 0u : f2key(m);
  const float M = block_max<NT>(m, sc);
  s = block_sum<NT>(m == -INFINITY ? 0.f : s * __expf(m - M), sc);
  const float lse = M + __logf(s);
  if (threadIdx.x == 0 && lse_out) lse_out[row] = lse;

  const int nonempty = min(NT, (V + 7) / 8);
  bool fast = K <= nonempty;
  int P2 = KP;
  if (fast) {
    // K-th largest of the thread maxima: radix select over 1024 keys
    uint32_t prefix = 0;
    int kth = K;
#pragma unroll 1
    for (int p = 0; p < NPASS; ++p) {
      const int shift = 24 - 8 * p;
      for (int i = threadIdx.x; i < 256; i += NT) hist[i] = 0;
      __syncthreads();
      const uint32_t k = tkeys[threadIdx.x];
      if (p == 0 || (k >> (shift + 8)) == prefix) atomicAdd(&hist[(k >> shift) & 255], 1);
      __syncthreads();
      pick_bin(hist, kth, &sel_bin, &sel_kth);
      __syncthreads();
      prefix = (prefix << 8) | (uint32_t)sel_bin;
      kth = sel_kth;
    }
    const uint32_t t0 = prefix << (32 - 8 * NPASS);  // low bits zero: a lower bound of the K-th thread max
    // ---- pass B: gather every element with key >= t0: only the chunks whose maximum reaches t0 are read again
    // (from L2 / Infinity Cache); rows too long for the register-resident maxima are re-read whole
    auto gather = [&](int c) {
      float f[8];
      load8<T>(x + c, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t k = f2key(f[e]);
        if (k >= t0) {
          const int slot = atomicAdd(&n_out, 1);
          if (slot < CAP) items[slot] = ((unsigned long long)k << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)(c + e));
        }
      }
    };
    if (keep_max) {
#pragma unroll
      for (int i = 0; i < MAXCH; ++i) {
        const int c = (i * NT + (int)threadIdx.x) * 8;
        if (c < V && f2key(cmx[i]) >= t0) gather(c);
      }
    } else {
      for (int c = threadIdx.x * 8; c < V; c += NT * 8) gather(c);
    }
    __syncthreads();
    const int nc = n_out;
    fast = nc <= CAP;
    if (fast) {
      P2 = 2;
      while (P2 < nc) P2 <<= 1;
      for (int i = nc + threadIdx.x; i < P2; i += NT) items[i] = 0ull;
      __syncthreads();
    } else {
      __syncthreads();
      if (threadIdx.x == 0) n_out = 0;
    }
  }
  if (!fast) {
    // ---- general path: MSB-first radix select over the whole row
    uint32_t prefix = 0;
    int kth = K;
#pragma unroll 1
    for (int p = 0; p < NPASS; ++p) {
      const int shift = 24 - 8 * p;
      for (int i = threadIdx.x; i < 256; i += NT) hist[i] = 0;
      __syncthreads();
      for (int c0 = 0; c0 < V; c0 += NT * 8) {
        const int c = c0 + threadIdx.x * 8;
        const bool in = c < V;
        float f[8];
        if (in) load8<T>(x + c, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t k = in ? f2key(f[e]) : 0u;
          hist_add(hist, (int)((k >> shift) & 255), in && (p == 0 || (k >> (shift + 8)) == prefix));
        }
      }
      __syncthreads();
      pick_bin(hist, kth, &sel_bin, &sel_kth);
      __syncthreads();
      prefix = (prefix << 8) | (uint32_t)sel_bin;
      kth = sel_kth;
    }
    const uint32_t thr = prefix << (32 - 8 * NPASS);
    const uint32_t lowmask = (NPASS == 4) ? 0u : ((1u << (32 - 8 * NPASS)) - 1u);
    const int need_ties = kth;
    // collection in index order: keys above the threshold always, the first `need_ties` ties
    for (int c0 = 0; c0 < V; c0 += NT * 8) {
      const int c = c0 + threadIdx.x * 8;
      const bool in = c < V;
      float f[8];
      uint32_t kk[8];
      int nt_mine = 0;
      if (in) load8<T>(x + c, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        kk[e] = in ? f2key(f[e]) : 0u;
        if (in && (kk[e] & ~lowmask) == thr) ++nt_mine;
      }
      int incl = nt_mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
      }
      if (lane == 63) wave_cnt[wv] = incl;
      __syncthreads();
      int wbase = tie_base, tot = 0;
      for (int w2 = 0; w2 < NT / 64; ++w2) {
        const int cw = wave_cnt[w2];
        if (w2 < wv) wbase += cw;
        tot += cw;
      }
      int rank = wbase + incl - nt_mine;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (!in) continue;
        const uint32_t hk = kk[e] & ~lowmask;
        bool take = hk > thr;
        if (hk == thr) { take = rank < need_ties; ++rank; }
        if (take) {
          const int slot = atomicAdd(&n_out, 1);
          if (slot < CAP) items[slot] = ((unsigned long long)kk[e] << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)(c + e));
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) tie_base += tot;
      __syncthreads();
    }
    P2 = KP;
    for (int i = threadIdx.x; i < P2; i += NT)
      if (i >= n_out || i >= K) items[i] = 0ull;
    __syncthreads();
  }
  bitonic_desc<NT>(items, P2);
  for (int i = threadIdx.x; i < K; i += NT) {
    const unsigned long long it = items[i];
    const float val = key2f((uint32_t)(it >> 32));
    outv[(long)row * K + i] = (_Float16)(val - lse);
    outi[(long)row * K + i] = (int32_t)(0xFFFFFFFFu - (uint32_t)it);
  }
}

}  // namespace

extern "C" int sd_logsoftmax_topk(const void* logits, void* top_v, void* top_i, float* lse_out, int rows,
                                  int64_t row_stride, int V, int K, int dtype, void* stream) {
  if (rows <= 0 || V <= 0 || K <= 0 || K > KMAX || K > V) return SD_ERR_SHAPE;
  if ((V & 7) || (row_stride & 7) || ((uintptr_t)logits & 15)) return SD_ERR_ALIGN;
  int KP = 2;
  while (KP < K) KP <<= 1;
  hipStream_t st = (hipStream_t)stream;
  SdProfScope prof(SD_K_TOPK, (double)rows * V * (dtype == SD_DTYPE_BF16 ? 2 : 4), st);
  const int nt_env = g_sd_debug.topk_nt;
  const int nt = (nt_env == 256 || nt_env == 512 || nt_env == 1024) ? nt_env : 512;
#define SD_TOPK_GO(T_, NP_, NT_)                                                                                     \
  SD_PROF_LABEL("topk_kernel<%s, %d, %d>", sizeof(T_) == 2 ? "__bf16" : "float", NP_, NT_);                          \
  hipLaunchKernelGGL((topk_kernel<T_, NP_, NT_>), dim3(rows), dim3(NT_), 0, st, (const T_*)logits, (_Float16*)top_v, \
                     (int32_t*)top_i, lse_out, rows, (long)row_stride, V, K, KP)
  if (dtype == SD_DTYPE_BF16) {
    if (nt == 256) { SD_TOPK_GO(bf16, 2, 256); } else if (nt == 1024) { SD_TOPK_GO(bf16, 2, 1024); } else { SD_TOPK_GO(bf16, 2, 512); }
  } else if (dtype == SD_DTYPE_F32) {
    if (nt == 256) { SD_TOPK_GO(float, 4, 256); } else if (nt == 1024) { SD_TOPK_GO(float, 4, 1024); } else { SD_TOPK_GO(float, 4, 512); }
  } else {
    return SD_ERR_UNSUPPORTED;
  }
#undef SD_TOPK_GO
  SD_CHECK_LAUNCH();
  return 0;
}
