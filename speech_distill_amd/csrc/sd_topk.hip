// Teacher log-softmax + exact top-K select over the vocabulary, one pass family per row, gfx950.
//
// Replaces /root/reference/train.py:80-91 (on-the-fly sparse extraction inside compute_loss) and
// /root/reference/extract_teacher_logits.py:114-129 (offline extraction):
//     logits[..., :V] -> log_softmax (T=1) -> topk(K), sorted descending -> values fp16, indices int32
// Selection is an exact MSB-first radix select on order-preserving integer keys of the raw logits
// (log-softmax is monotone, so selecting on logits == selecting on log-probs); ties at the K-th
// value go to the LOWEST indices.  One 1024-thread workgroup per row; the row is re-read from
// L2/Infinity Cache by the later radix passes, so HBM sees it about once.
#include "sd_common.cuh"
#include "../../include/sd_hip.h"

namespace {

constexpr int NT = 1024;
constexpr int KMAX = 1024;

SD_DEV uint32_t f2key(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
SD_DEV float key2f(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
  return __uint_as_float(u);
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

template <typename T, int NPASS>
__global__ __launch_bounds__(NT) void topk_kernel(const T* __restrict__ X, _Float16* __restrict__ outv,
                                                  int32_t* __restrict__ outi, float* __restrict__ lse_out, int rows,
                                                  long row_stride, int V, int K, int KP) {
  __shared__ float sc[32];
  __shared__ int hist[256];
  __shared__ int sel_bin, sel_kth, n_out, tie_base;
  __shared__ int wave_cnt[16];
  __shared__ unsigned long long items[KMAX];
  const int row = blockIdx.x;
  const T* x = X + (long)row * row_stride;
  const int lane = lane_id(), wv = threadIdx.x >> 6;

  // ---- pass 0: log-sum-exp statistics + histogram of the top key byte
  for (int i = threadIdx.x; i < 256; i += NT) hist[i] = 0;
  if (threadIdx.x == 0) { n_out = 0; tie_base = 0; }
  __syncthreads();
  float m = -INFINITY, s = 0.f;
  for (int c0 = 0; c0 < V; c0 += NT * 8) {
    const int c = c0 + threadIdx.x * 8;
    const bool in = c < V;
    float f[8];
    if (in) load8<T>(x + c, f);
    if (in) {
      float cm = f[0];
#pragma unroll
      for (int e = 1; e < 8; ++e) cm = fmaxf(cm, f[e]);
      if (cm > m) { s *= __expf(m - cm); m = cm; }
#pragma unroll
      for (int e = 0; e < 8; ++e) s += __expf(f[e] - m);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) hist_add(hist, in ? (int)(f2key(f[e]) >> 24) : 0, in);
  }
  const float M = block_max<NT>(m, sc);
  s = block_sum<NT>(m == -INFINITY ? 0.f : s * __expf(m - M), sc);
  const float lse = M + __logf(s);
  if (threadIdx.x == 0 && lse_out) lse_out[row] = lse;
  __syncthreads();
  pick_bin(hist, K, &sel_bin, &sel_kth);
  __syncthreads();
  uint32_t prefix = (uint32_t)sel_bin;
  int kth = sel_kth;

  // ---- passes 1..NPASS-1: refine inside the selected bin
#pragma unroll 1
  for (int p = 1; p < NPASS; ++p) {
    const int shift = 24 - 8 * p;
    for (int i = threadIdx.x; i < 256; i += NT) hist[i] = 0;
    __syncthreads();
    for (int c = threadIdx.x * 8; c < V; c += NT * 8) {
      float f[8];
      load8<T>(x + c, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t k = f2key(f[e]);
        if ((k >> (shift + 8)) == prefix) atomicAdd(&hist[(k >> shift) & 255], 1);
      }
    }
    __syncthreads();
    pick_bin(hist, kth, &sel_bin, &sel_kth);
    __syncthreads();
    prefix = (prefix << 8) | (uint32_t)sel_bin;
    kth = sel_kth;
  }
  // threshold key: top 8*NPASS bits = prefix.  `kth` ties at the threshold are still needed.
  const uint32_t thr = prefix << (32 - 8 * NPASS);
  const uint32_t lowmask = (NPASS == 4) ? 0u : ((1u << (32 - 8 * NPASS)) - 1u);
  const int need_ties = kth;

  // ---- collection, in index order: keys above the threshold always, the first `need_ties` ties
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
      const uint32_t hk = kk[e] & ~lowmask;
      if (in && hk == thr) ++nt_mine;
    }
    // block-wide exclusive scan of the tie counts (index order = thread order within this sweep)
    int incl = nt_mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    if (lane == 63) wave_cnt[wv] = incl;
    __syncthreads();
    int wbase = tie_base, tot = 0;
    for (int w2 = 0; w2 < 16; ++w2) {
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
        if (slot < KMAX) items[slot] = ((unsigned long long)kk[e] << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)(c + e));
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) tie_base += tot;
    __syncthreads();
  }
  // ---- bitonic sort (descending) of KP >= K composite keys; value first, lower index first on ties
  for (int i = threadIdx.x; i < KP; i += NT)
    if (i >= n_out || i >= K) items[i] = 0ull;
  __syncthreads();
  for (int size = 2; size <= KP; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = threadIdx.x; i < KP / 2; i += NT) {
        const int lo = ((i / stride) * stride * 2) + (i % stride);
        const int hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long a = items[lo], b = items[hi];
        if (desc ? (a < b) : (a > b)) { items[lo] = b; items[hi] = a; }
      }
      __syncthreads();
    }
  }
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
  if (dtype == SD_DTYPE_BF16)
    hipLaunchKernelGGL((topk_kernel<bf16, 2>), dim3(rows), dim3(NT), 0, st, (const bf16*)logits, (_Float16*)top_v,
                       (int32_t*)top_i, lse_out, rows, (long)row_stride, V, K, KP);
  else if (dtype == SD_DTYPE_F32)
    hipLaunchKernelGGL((topk_kernel<float, 4>), dim3(rows), dim3(NT), 0, st, (const float*)logits, (_Float16*)top_v,
                       (int32_t*)top_i, lse_out, rows, (long)row_stride, V, K, KP);
  else
    return SD_ERR_UNSUPPORTED;
  SD_CHECK_LAUNCH();
  return 0;
}
