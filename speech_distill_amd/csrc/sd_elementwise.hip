// HBM-bound row kernels of the Qwen3 decoder for gfx950: RMSNorm fwd/bwd, per-head q/k RMSNorm +
// rotate-half RoPE fwd/bwd, SwiGLU fwd/bwd, embedding gather / deterministic scatter-add.
//
// Arithmetic restated from the third-party HF module the reference calls (train.py:54, 63-69):
//   RMSNorm ........ modeling_qwen3.py:59-64  (fp32 statistics, cast to bf16, THEN * gain)
//   q/k norm ....... modeling_qwen3.py:252-253 (RMSNorm over head_dim=128, before RoPE)
//   RoPE ........... modeling_qwen3.py:121-170 (theta^(-2i/d), cat(freqs,freqs), rotate-half;
//                    cos/sin are rounded to the activation dtype before use)
//   SwiGLU ......... modeling_qwen3.py:81-83
//   embedding ...... modeling_qwen3.py:381
// Every kernel moves 16 bytes per lane per access (8 bf16) and reduces with wave shuffles.
#include <stdlib.h>
#include "sd_common.cuh"
#include "../../include/sd_hip.h"
#include "sd_prof.h"
#include "sd_debug.h"

namespace {

SD_DEV void unpack8(bf16x8 v, float* f) {
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
}
SD_DEV bf16x8 pack8(const float* f) {
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (bf16)f[e];
  return v;
}

// ------------------------------------------------------------------------------------------ RMSNorm
// One wave per row, the row held in registers (NCH chunks of 512 columns) between the statistic and
// the output pass: one HBM read, one write.
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                          bf16* __restrict__ y, float* __restrict__ rstd_out, int M,
                                                          int H, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = lane_id();
  const bf16* xr = x + (long)row * H;
  bf16x8 xv[NCH];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane * 8 + i * 512;
    if (c < H) {
      xv[i] = *(const bf16x8*)(xr + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float f = (float)xv[i][e]; ss += f * f; }
    }
  }
  ss = wave_sum(ss);
  const float rstd = rsqrtf(ss / (float)H + eps);
  if (lane == 0 && rstd_out) rstd_out[row] = rstd;
  bf16* yr = y + (long)row * H;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane * 8 + i * 512;
    if (c < H) {
      float f[8], g[8];
      unpack8(xv[i], f);
      unpack8(*(const bf16x8*)(w + c), g);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = g[e] * (float)(bf16)(f[e] * rstd);  // HF: weight * hidden.to(bf16)
      *(bf16x8*)(yr + c) = pack8(f);
    }
  }
}

// dx = rstd * (g - xhat * mean(g*xhat)),  g = dy*w, xhat = x*rstd;  dx (+)= dres.
// Gain gradient: every lane owns fixed columns and keeps their sums over this block's rows in
// registers; the 4 waves are combined through LDS in a fixed order (deterministic), one partial
// row per block, summed by colsum_reduce_kernel.
// SLABS: dy is given as `nsplit` fp32 slabs [nsplit][M][H] (the un-reduced output of a split-K GEMM), summed here
// in slab order -- the separate split-K reduce pass and its bf16 rounding disappear.
template <int NCH, bool SLABS>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16* __restrict__ dy, const float* __restrict__ dy_slabs,
                                                          int nsplit, const bf16* __restrict__ x,
                                                          const bf16* __restrict__ w, const float* __restrict__ rstd,
                                                          const bf16* dres, bf16* dx, float* __restrict__ dw_part,
                                                          int M, int H, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float dw_s[];  // [4][H]
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float acc[NCH][8];
  float wf[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane * 8 + i * 512;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
    if (c < H) unpack8(*(const bf16x8*)(w + c), wf[i]);
  }
  for (int row = r0 + wv; row < r1; row += 4) {
    const bf16* xr = x + (long)row * H;
    const bf16* dyr = SLABS ? nullptr : dy + (long)row * H;
    const float rs = rstd[row];
    bf16x8 xv[NCH];
    float dv[NCH][8];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane * 8 + i * 512;
      if (c < H) {
        xv[i] = *(const bf16x8*)(xr + c);
        if constexpr (SLABS) {
#pragma unroll
          for (int e = 0; e < 8; ++e) dv[i][e] = 0.f;
          for (int sp = 0; sp < nsplit; ++sp) {
            const float* ps = dy_slabs + ((long)sp * M + row) * H + c;
            const f32x4 a = *(const f32x4*)ps, b = *(const f32x4*)(ps + 4);
            dv[i][0] += a[0]; dv[i][1] += a[1]; dv[i][2] += a[2]; dv[i][3] += a[3];
            dv[i][4] += b[0]; dv[i][5] += b[1]; dv[i][6] += b[2]; dv[i][7] += b[3];
          }
        } else {
          unpack8(*(const bf16x8*)(dyr + c), dv[i]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) dot += dv[i][e] * wf[i][e] * (float)xv[i][e] * rs;
      }
    }
    dot = wave_sum(dot) / (float)H;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane * 8 + i * 512;
      if (c < H) {
        float o[8];
        if (dres) unpack8(*(const bf16x8*)(dres + (long)row * H + c), o);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (float)xv[i][e] * rs, d = dv[i][e];
          const float v = rs * (d * wf[i][e] - xh * dot);
          o[e] = dres ? o[e] + v : v;
          acc[i][e] += d * xh;
        }
        *(bf16x8*)(dx + (long)row * H + c) = pack8(o);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane * 8 + i * 512;
    if (c < H)
#pragma unroll
      for (int e = 0; e < 8; ++e) dw_s[wv * H + c + e] = acc[i][e];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 256)
    dw_part[(long)blockIdx.x * H + c] = (dw_s[c] + dw_s[H + c]) + (dw_s[2 * H + c] + dw_s[3 * H + c]);
}

// out[c] (bf16) = (accumulate ? out[c] : 0) + sum_b part[b*stride + c].  Block = 32 columns x 8 row
// groups; every thread sums its rows with 8 independent loads in flight, then a fixed-order LDS reduce.
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ part, bf16* out, int nb, int H,
                                                            int stride, int accumulate) {
  __shared__ float red[8][32];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (c < H) {
    int b = rg;
    for (; b + 56 < nb; b += 64) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = part[(long)(b + 8 * u) * stride + c];
      s += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
    for (; b < nb; b += 8) s += part[(long)b * stride + c];
  }
  red[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && c < H) {
    float t = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) + ((red[4][cl] + red[5][cl]) + (red[6][cl] + red[7][cl]));
    if (accumulate) t += (float)out[c];
    out[c] = (bf16)t;
  }
}

// Up to 8 independent column sums in ONE launch (a layer's gain gradients: input norm, post-attention norm, q gain,
// k gain -- four launches of ~5 us each otherwise).  Block -> problem by a prefix table of 32-column blocks.
struct ColsumBatch {
  sd_colsum_problem p[8];
  int first_block[9];
  int n;
};
__global__ __launch_bounds__(256) void colsum_reduce_batch_kernel(ColsumBatch b) {
  __shared__ float red[8][32];
  int pi = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i)
    if (i < b.n && (int)blockIdx.x >= b.first_block[i]) pi = i;
  const sd_colsum_problem q = b.p[pi];
  const float* part = q.partials;
  bf16* out = (bf16*)q.out;
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = ((int)blockIdx.x - b.first_block[pi]) * 32 + cl;
  float s = 0.f;
  if (c < q.H) {
    int r = rg;
    for (; r + 56 < q.nb; r += 64) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = part[(long)(r + 8 * u) * q.stride + c];
      s += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
    for (; r < q.nb; r += 8) s += part[(long)r * q.stride + c];
  }
  red[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && c < q.H) {
    float t = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) + ((red[4][cl] + red[5][cl]) + (red[6][cl] + red[7][cl]));
    if (q.accumulate) t += (float)out[c];
    out[c] = (bf16)t;
  }
}

// ---------------------------------------------------------------------------- q/k norm + RoPE (d=128)
// A head vector of 128 bf16 is owned by 16 lanes (8 elements each); a wave does 4 heads per step.
// rotate_half partner of element i is i^64  <->  lane j ^ 8 inside the 16-lane group.
__global__ __launch_bounds__(256) void qknorm_rope_fwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ qw,
                                                              const bf16* __restrict__ kw, const bf16* __restrict__ cosb,
                                                              const bf16* __restrict__ sinb, bf16* __restrict__ out,
                                                              int M, int T, int Hq, int Hkv, float eps) {
  const int lane = lane_id();
  const int sub = lane >> 4, j = lane & 15;
  const int nh = Hq + Hkv;
  const long total = (long)M * nh;
  const long idx = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + sub;
  const bool ok = idx < total;
  const long id = ok ? idx : total - 1;
  const int m = (int)(id / nh), hh = (int)(id % nh);
  const int t = m % T;
  const bf16* src = qkv + (long)m * (Hq + 2 * Hkv) * 128 + hh * 128 + j * 8;
  float f[8], g[8], cs[8], sn[8];
  unpack8(*(const bf16x8*)src, f);
  float ss = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) ss += f[e] * f[e];
  ss = row16_sum(ss);
  const float rs = rsqrtf(ss * (1.f / 128.f) + eps);
  unpack8(*(const bf16x8*)((hh < Hq ? qw : kw) + j * 8), g);
  unpack8(*(const bf16x8*)(cosb + (long)t * 128 + j * 8), cs);
  unpack8(*(const bf16x8*)(sinb + (long)t * 128 + j * 8), sn);
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float n = (float)(bf16)(g[e] * (float)(bf16)(f[e] * rs));  // normed value as HF holds it (bf16)
    const float p = row16_xor8(n);
    const float rot = (j < 8) ? -p : p;
    o[e] = n * cs[e] + rot * sn[e];
  }
  if (ok) *(bf16x8*)(out + (long)m * nh * 128 + hh * 128 + j * 8) = pack8(o);
}

// Backward of the above.  dout [M,(Hq+Hkv)*128] -> dqkv q/k slots [M,(Hq+2Hkv)*128]; gain grads as
// per-block partial sums [nblk][256] (q gain in [0,128), k gain in [128,256)).
__global__ __launch_bounds__(256) void qknorm_rope_bwd_kernel(const bf16* __restrict__ dout, const bf16* __restrict__ qkv,
                                                              const bf16* __restrict__ qw, const bf16* __restrict__ kw,
                                                              const bf16* __restrict__ cosb, const bf16* __restrict__ sinb,
                                                              bf16* __restrict__ dqkv, float* __restrict__ dw_part, int M,
                                                              int T, int Hq, int Hkv, float eps, int items_per_block) {
  __shared__ float dw_s[16][256];  // [wave*4+sub][q gain 0..127 | k gain 128..255]
  const int lane = lane_id();
  const int sub = lane >> 4, j = lane & 15;
  const int nh = Hq + Hkv;
  const long total = (long)M * nh;
  const long i0 = (long)blockIdx.x * items_per_block;
  float accw[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // this lane's gain-gradient partial for q (hh<Hq) ...
  float acck[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // ... and for k
  const bf16x8 gq = *(const bf16x8*)(qw + j * 8), gk = *(const bf16x8*)(kw + j * 8);
  // UN items per 16-lane group per trip, every load of the trip issued before the first use: the kernel is a chain of
  // dependent HBM round trips otherwise (one item = 5 loads -> shuffles -> 1 store, 24 us for 38 MB at UN = 1)
  constexpr int UN = 3;
  for (long base = i0 + (threadIdx.x >> 6) * 4; base < i0 + items_per_block; base += 16 * UN) {
    bf16x8 fv[UN], dyv[UN], csv[UN], snv[UN];
    long ids[UN];
    bool oks[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long idx = base + 16 * u + sub;
      oks[u] = idx < total && idx < i0 + items_per_block;
      ids[u] = idx < total ? idx : total - 1;
      const int m = (int)(ids[u] / nh), hh = (int)(ids[u] % nh);
      const int t = m % T;
      fv[u] = *(const bf16x8*)(qkv + (long)m * (Hq + 2 * Hkv) * 128 + hh * 128 + j * 8);
      dyv[u] = *(const bf16x8*)(dout + (long)m * nh * 128 + hh * 128 + j * 8);
      csv[u] = *(const bf16x8*)(cosb + (long)t * 128 + j * 8);
      snv[u] = *(const bf16x8*)(sinb + (long)t * 128 + j * 8);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const bool ok = oks[u];
      const int m = (int)(ids[u] / nh), hh = (int)(ids[u] % nh);
      float f[8], g[8], cs[8], sn[8], dy[8];
      unpack8(fv[u], f);
      unpack8(dyv[u], dy);
      unpack8(hh < Hq ? gq : gk, g);
      unpack8(csv[u], cs);
      unpack8(snv[u], sn);
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += f[e] * f[e];
      ss = row16_sum(ss);
      const float rs = rsqrtf(ss * (1.f / 128.f) + eps);
      // RoPE^T:  dn[i] = dy[i] cos[i] + (i<64 ?  dy[i+64] sin[i+64] : -dy[i-64] sin[i-64])
      float dn[8], dot = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float z = dy[e] * sn[e];
        const float pz = row16_xor8(z);
        dn[e] = dy[e] * cs[e] + ((j < 8) ? pz : -pz);
        dot += dn[e] * g[e] * f[e] * rs;
      }
      dot = row16_sum(dot);
      dot *= (1.f / 128.f);
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = f[e] * rs;
        o[e] = rs * (dn[e] * g[e] - xh * dot);
        const float gw = ok ? dn[e] * xh : 0.f;
        if (hh < Hq) accw[e] += gw; else acck[e] += gw;
      }
      if (ok) *(bf16x8*)(dqkv + (long)m * (Hq + 2 * Hkv) * 128 + hh * 128 + j * 8) = pack8(o);
    }
  }
  const int slot = (threadIdx.x >> 6) * 4 + sub;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    dw_s[slot][j * 8 + e] = accw[e];
    dw_s[slot][128 + j * 8 + e] = acck[e];
  }
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) t += dw_s[q][threadIdx.x];
  dw_part[(long)blockIdx.x * 256 + threadIdx.x] = t;
}

// ------------------------------------------------------------------------------------------- SwiGLU
// gu [M, 2I]: gate in columns [0,I), up in [I,2I).  act [M, I] = silu(gate) * up.
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16* __restrict__ gu, bf16* __restrict__ act, long n8,
                                                         int I) {
  const int i8 = I >> 3;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n8; q += (long)gridDim.x * 256) {
    const long m = q / i8;
    const int c = (int)(q % i8) * 8;
    float g[8], u[8], o[8];
    unpack8(*(const bf16x8*)(gu + m * 2 * I + c), g);
    unpack8(*(const bf16x8*)(gu + m * 2 * I + I + c), u);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = g[e] / (1.f + __expf(-g[e])) * u[e];
    *(bf16x8*)(act + m * I + c) = pack8(o);
  }
}
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16* __restrict__ dact, const bf16* __restrict__ gu,
                                                         bf16* __restrict__ dgu, long n8, int I) {
  const int i8 = I >> 3;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n8; q += (long)gridDim.x * 256) {
    const long m = q / i8;
    const int c = (int)(q % i8) * 8;
    float g[8], u[8], d[8], dg[8], du[8];
    unpack8(*(const bf16x8*)(gu + m * 2 * I + c), g);
    unpack8(*(const bf16x8*)(gu + m * 2 * I + I + c), u);
    unpack8(*(const bf16x8*)(dact + m * I + c), d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float s = 1.f / (1.f + __expf(-g[e]));
      du[e] = d[e] * g[e] * s;
      dg[e] = d[e] * u[e] * s * (1.f + g[e] * (1.f - s));
    }
    *(bf16x8*)(dgu + m * 2 * I + c) = pack8(dg);
    *(bf16x8*)(dgu + m * 2 * I + I + c) = pack8(du);
  }
}

// ---------------------------------------------------------------------------------------- embedding
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const int64_t* __restrict__ ids, const bf16* __restrict__ E,
                                                            bf16* __restrict__ x, int M, int H, int V) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  long id = ids[row];
  id = id < 0 ? 0 : (id >= V ? V - 1 : id);
  for (int c = lane_id() * 8; c < H; c += 512) *(bf16x8*)(x + (long)row * H + c) = *(const bf16x8*)(E + id * H + c);
}

// The same lookup, leaving per row and 128-column tile the sum of squares of the row (fp32, tile-major [H/128][M]): the statistic of the
// FIRST decoder layer's input norm when that norm is folded into the q|k|v projection (sd_gemm_qkv_rope_rs).  A lane's 8
// columns of chunk i lie in tile (lane >> 4) + 4 i, so one 16-lane DPP row sum is one tile.
__global__ __launch_bounds__(256) void embedding_fwd_ssq_kernel(const int64_t* __restrict__ ids, const bf16* __restrict__ E,
                                                                bf16* __restrict__ x, float* __restrict__ ssq, int M, int H,
                                                                int V) {
  // a workgroup takes 16 consecutive rows (a wave 4 of them): the 64-byte lines of the tile-major ssq [H/128][M] it
  // writes are its own
  const int lane = lane_id();
  for (int rr = 0; rr < 4; ++rr) {
    const int row = blockIdx.x * 16 + (threadIdx.x >> 6) * 4 + rr;
    if (row >= M) return;  // wave-uniform
    long id = ids[row];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    for (int i = 0; i * 512 < H; ++i) {  // H % 512 == 0: every lane is active in every trip (DPP needs the full row)
      const int c = lane * 8 + i * 512;
      const bf16x8 v = *(const bf16x8*)(E + id * H + c);
      *(bf16x8*)(x + (long)row * H + c) = v;
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; ss += f * f; }
      ss = row16_sum(ss);
      if ((lane & 15) == 0) ssq[(long)((lane >> 4) + 4 * i) * M + row] = ss;
    }
  }
}

// dst[rows[i]] = src[i] for unique row indices (dst was zero-filled by the launcher).
__global__ __launch_bounds__(256) void rows_scatter_kernel(const bf16* __restrict__ src, const int64_t* __restrict__ rows,
                                                           bf16* __restrict__ dst, int n, int M, int H) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const long r = rows[i];
  if (r < 0 || r >= M) return;
  for (int c = lane_id() * 8; c < H; c += 512) *(bf16x8*)(dst + r * H + c) = *(const bf16x8*)(src + (long)i * H + c);
}

// Deterministic scatter-add: the block of the FIRST token carrying an id sums every token row with
// that id (fixed order) and adds the total to dE[id]; other blocks exit.  No atomics on the gradient, no sort.
// The later tokens with the same id are found by all 256 threads at once (a bitmap in LDS, walked in increasing token
// order afterwards) -- every thread scanning all M ids itself cost 83 us at M = 2 048 and grew with M squared.
constexpr int EMB_BM_WORDS = 2048;  // bitmap for up to 65 536 tokens; longer batches take the serial scan
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int64_t* __restrict__ ids, const bf16* __restrict__ dx,
                                                            bf16* dE, int M, int H, int V, float scale) {
  __shared__ int first_flag;
  __shared__ unsigned same[EMB_BM_WORDS];
  const int m = blockIdx.x;
  const long id = ids[m];
  if (id < 0 || id >= V) return;
  if (threadIdx.x == 0) first_flag = 1;
  __syncthreads();
  for (int i = threadIdx.x; i < m; i += 256)
    if (ids[i] == id) first_flag = 0;
  __syncthreads();
  if (!first_flag) return;
  const int nwords = (M + 31) >> 5;
  const bool mapped = nwords <= EMB_BM_WORDS;
  if (mapped) {
    for (int w = (m >> 5) + threadIdx.x; w < nwords; w += 256) same[w] = 0u;
    __syncthreads();
    for (int i = m + threadIdx.x; i < M; i += 256)
      if (ids[i] == id) atomicOr(&same[i >> 5], 1u << (i & 31));
    __syncthreads();
  }
  for (int c = threadIdx.x * 8; c < H; c += 2048) {
    float acc[8], f[8], base[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (mapped) {
      for (int w = m >> 5; w < nwords; ++w) {
        unsigned bits = same[w];
        while (bits) {  // increasing token order: the sum is the same whatever the launch looked like
          const int i = (w << 5) + __builtin_ctz(bits);
          bits &= bits - 1;
          unpack8(*(const bf16x8*)(dx + (long)i * H + c), f);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += f[e];
        }
      }
    } else {
      for (int i = m; i < M; ++i) {
        if (ids[i] != id) continue;
        unpack8(*(const bf16x8*)(dx + (long)i * H + c), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += f[e];
      }
    }
    unpack8(*(const bf16x8*)(dE + id * H + c), base);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = base[e] + scale * acc[e];
    *(bf16x8*)(dE + id * H + c) = pack8(acc);
  }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int sd_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int M, int H, float eps, void* stream) {
  if (M <= 0 || (H & 7)) return SD_ERR_SHAPE;
  if (H > 4096) return SD_ERR_UNSUPPORTED;
  SdProfScope prof(SD_K_RMSNORM, 4.0 * M * H, ST);
  SD_PROF_LABEL("rmsnorm_fwd_kernel<%d>", (H + 511) / 512 <= 1 ? 1 : (H + 511) / 512 == 2 ? 2 : (H + 511) / 512 <= 4 ? 4 : 8);
#define SD_RMS_FWD(N) hipLaunchKernelGGL(rmsnorm_fwd_kernel<N>, dim3((M + 3) / 4), dim3(256), 0, ST, (const bf16*)x, \
                                         (const bf16*)w, (bf16*)y, rstd, M, H, eps)
  const int nch = (H + 511) / 512;
  if (nch <= 1) SD_RMS_FWD(1); else if (nch == 2) SD_RMS_FWD(2); else if (nch <= 4) SD_RMS_FWD(4); else SD_RMS_FWD(8);
#undef SD_RMS_FWD
  SD_CHECK_LAUNCH();
  return 0;
}

extern "C" int64_t sd_rmsnorm_bwd_workspace_bytes(int M, int H) {
  const int nb = (M + 3) / 4 < 512 ? (M + 3) / 4 : 512;
  return (int64_t)nb * H * 4;
}

// reduce_stream / event (both nullable): the gain-gradient reduce only feeds the optimizer, so it may run on a
// second stream beside the dX chain; `event` orders it after the partials.  The caller joins the streams.
static int rmsnorm_bwd_any(const void* dy, const float* dy_slabs, int nsplit, const void* x, const void* w,
                           const float* rstd, const void* dres, void* dx, void* dw, int accumulate_dw, void* workspace,
                           int M, int H, void* reduce_stream, void* event, void* stream) {
  if (M <= 0 || (H & 7)) return SD_ERR_SHAPE;
  if (H > 4096) return SD_ERR_UNSUPPORTED;
  int nb = (M + 3) / 4 < 512 ? (M + 3) / 4 : 512;  // 2 workgroups (8 waves) per CU, one or two rows per wave
  const int rpb = (M + nb - 1) / nb;
  nb = (M + rpb - 1) / rpb;
  SdProfScope prof(SD_K_RMSNORM, ((dres ? 8.0 : 6.0) + (dy_slabs ? 4.0 * nsplit - 2.0 : 0.0)) * M * H, ST);
  SD_PROF_LABEL("rmsnorm_bwd_kernel<%d, %s>", (H + 511) / 512 <= 1 ? 1 : (H + 511) / 512 == 2 ? 2 : (H + 511) / 512 <= 4 ? 4 : 8, dy_slabs ? "true" : "false");
#define SD_RMS_BWD(N)                                                                                                  \
  do {                                                                                                                 \
    if (dy_slabs)                                                                                                      \
      hipLaunchKernelGGL((rmsnorm_bwd_kernel<N, true>), dim3(nb), dim3(256), 4 * H * 4, ST, (const bf16*)nullptr,       \
                         dy_slabs, nsplit, (const bf16*)x, (const bf16*)w, rstd, (const bf16*)dres, (bf16*)dx,           \
                         (float*)workspace, M, H, rpb);                                                                \
    else                                                                                                               \
      hipLaunchKernelGGL((rmsnorm_bwd_kernel<N, false>), dim3(nb), dim3(256), 4 * H * 4, ST, (const bf16*)dy,           \
                         (const float*)nullptr, 0, (const bf16*)x, (const bf16*)w, rstd, (const bf16*)dres, (bf16*)dx,   \
                         (float*)workspace, M, H, rpb);                                                                \
  } while (0)
  const int nch = (H + 511) / 512;
  if (nch <= 1) SD_RMS_BWD(1); else if (nch == 2) SD_RMS_BWD(2); else if (nch <= 4) SD_RMS_BWD(4); else SD_RMS_BWD(8);
#undef SD_RMS_BWD
  SD_CHECK_LAUNCH();
  if (!dw) return 0;  // partial sums stay in `workspace` ([sd_rmsnorm_bwd_partial_rows][H]); the caller reduces them
  hipStream_t rs = ST;
  if (reduce_stream && event) {
    rs = (hipStream_t)reduce_stream;
    if (hipEventRecord((hipEvent_t)event, ST) != hipSuccess || hipStreamWaitEvent(rs, (hipEvent_t)event, 0) != hipSuccess)
      return SD_ERR_WORKSPACE;
  }
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((H + 31) / 32), dim3(256), 0, rs, (const float*)workspace, (bf16*)dw,
                     nb, H, H, accumulate_dw);
  SD_CHECK_LAUNCH();
  return 0;
}

extern "C" int sd_rmsnorm_bwd_partial_rows(int M, int H) {
  (void)H;
  int nb = (M + 3) / 4 < 512 ? (M + 3) / 4 : 512;
  const int rpb = (M + nb - 1) / nb;
  return (M + rpb - 1) / rpb;
}

extern "C" int sd_colsum_reduce_batch(const sd_colsum_problem* problems_host, int n, void* stream) {
  if (n <= 0 || n > 8 || !problems_host) return SD_ERR_SHAPE;
  ColsumBatch b;
  b.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const sd_colsum_problem& q = problems_host[i];
    if (!q.partials || !q.out || q.nb <= 0 || q.H <= 0 || q.stride < q.H) return SD_ERR_SHAPE;
    b.p[i] = q;
    b.first_block[i] = blocks;
    blocks += (q.H + 31) / 32;
  }
  for (int i = n; i < 9; ++i) b.first_block[i] = blocks;
  for (int i = n; i < 8; ++i) b.p[i] = b.p[0];
  hipLaunchKernelGGL(colsum_reduce_batch_kernel, dim3(blocks), dim3(256), 0, ST, b);
  SD_CHECK_LAUNCH();
  return 0;
}

extern "C" int sd_rmsnorm_bwd2(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx,
                               void* dw, int accumulate_dw, void* workspace, int M, int H, void* reduce_stream,
                               void* event, void* stream) {
  return rmsnorm_bwd_any(dy, nullptr, 0, x, w, rstd, dres, dx, dw, accumulate_dw, workspace, M, H, reduce_stream, event,
                         stream);
}

extern "C" int sd_rmsnorm_bwd_slabs(const float* dy_slabs, int nsplit, const void* x, const void* w, const float* rstd,
                                    const void* dres, void* dx, void* dw, int accumulate_dw, void* workspace, int M, int H,
                                    void* reduce_stream, void* event, void* stream) {
  if (!dy_slabs || nsplit < 1 || ((uintptr_t)dy_slabs & 15)) return SD_ERR_SHAPE;
  return rmsnorm_bwd_any(nullptr, dy_slabs, nsplit, x, w, rstd, dres, dx, dw, accumulate_dw, workspace, M, H, reduce_stream,
                         event, stream);
}

extern "C" int sd_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx,
                              void* dw, int accumulate_dw, void* workspace, int M, int H, void* stream) {
  return sd_rmsnorm_bwd2(dy, x, w, rstd, dres, dx, dw, accumulate_dw, workspace, M, H, nullptr, nullptr, stream);
}

extern "C" int sd_qknorm_rope_fwd(const void* qkv, const void* q_gain, const void* k_gain, const void* cos_tab,
                                  const void* sin_tab, void* qk_out, int M, int T, int Hq, int Hkv, float eps,
                                  void* stream) {
  if (M <= 0 || T <= 0 || (M % T)) return SD_ERR_SHAPE;
  const long items = (long)M * (Hq + Hkv);
  SdProfScope prof(SD_K_QKROPE, 4.0 * items * 128, ST);
  SD_PROF_LABEL("qknorm_rope_fwd_kernel");
  hipLaunchKernelGGL(qknorm_rope_fwd_kernel, dim3((unsigned)((items + 15) / 16)), dim3(256), 0, ST, (const bf16*)qkv,
                     (const bf16*)q_gain, (const bf16*)k_gain, (const bf16*)cos_tab, (const bf16*)sin_tab, (bf16*)qk_out,
                     M, T, Hq, Hkv, eps);
  SD_CHECK_LAUNCH();
  return 0;
}

static inline int qk_bwd_blocks(long items, int* ipb) {
  const long nblk = g_sd_debug.qk_bwd_blocks > 0 ? g_sd_debug.qk_bwd_blocks : 512;
  long per = (items + nblk - 1) / nblk;
  per = (per + 47) / 48 * 48;  // a whole number of 16-lane-group trips of 3 items (qknorm_rope_bwd_kernel)
  *ipb = (int)per;
  return (int)((items + per - 1) / per);
}

extern "C" int64_t sd_qknorm_rope_bwd_workspace_bytes(int M, int Hq, int Hkv) {
  int ipb;
  return (int64_t)qk_bwd_blocks((long)M * (Hq + Hkv), &ipb) * 256 * 4;
}

extern "C" int sd_qknorm_rope_bwd_partial_rows(int M, int Hq, int Hkv) {
  int ipb;
  return qk_bwd_blocks((long)M * (Hq + Hkv), &ipb);
}

extern "C" int sd_qknorm_rope_bwd2(const void* dqk, const void* qkv, const void* q_gain, const void* k_gain,
                                   const void* cos_tab, const void* sin_tab, void* dqkv, void* dq_gain, void* dk_gain,
                                   int accumulate_dw, void* workspace, int M, int T, int Hq, int Hkv, float eps,
                                   void* reduce_stream, void* event, void* stream) {
  if (M <= 0 || T <= 0 || (M % T)) return SD_ERR_SHAPE;
  int ipb;
  const int nb = qk_bwd_blocks((long)M * (Hq + Hkv), &ipb);
  SdProfScope prof(SD_K_QKROPE, 6.0 * M * (Hq + Hkv) * 128, ST);
  SD_PROF_LABEL("qknorm_rope_bwd_kernel");
  hipLaunchKernelGGL(qknorm_rope_bwd_kernel, dim3(nb), dim3(256), 0, ST, (const bf16*)dqk, (const bf16*)qkv,
                     (const bf16*)q_gain, (const bf16*)k_gain, (const bf16*)cos_tab, (const bf16*)sin_tab, (bf16*)dqkv,
                     (float*)workspace, M, T, Hq, Hkv, eps, ipb);
  SD_CHECK_LAUNCH();
  // partial layout [nb][256]: columns 0..127 -> q gain, 128..255 -> k gain
  if (!dq_gain || !dk_gain) return 0;  // left in `workspace` ([sd_qknorm_rope_bwd_partial_rows][256]) for the caller
  hipStream_t rs = ST;
  if (reduce_stream && event) {
    rs = (hipStream_t)reduce_stream;
    if (hipEventRecord((hipEvent_t)event, ST) != hipSuccess || hipStreamWaitEvent(rs, (hipEvent_t)event, 0) != hipSuccess)
      return SD_ERR_WORKSPACE;
  }
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3(4), dim3(256), 0, rs, (const float*)workspace, (bf16*)dq_gain, nb, 128,
                     256, accumulate_dw);
  SD_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3(4), dim3(256), 0, rs, (const float*)workspace + 128, (bf16*)dk_gain,
                     nb, 128, 256, accumulate_dw);
  SD_CHECK_LAUNCH();
  return 0;
}

extern "C" int sd_qknorm_rope_bwd(const void* dqk, const void* qkv, const void* q_gain, const void* k_gain,
                                  const void* cos_tab, const void* sin_tab, void* dqkv, void* dq_gain, void* dk_gain,
                                  int accumulate_dw, void* workspace, int M, int T, int Hq, int Hkv, float eps,
                                  void* stream) {
  return sd_qknorm_rope_bwd2(dqk, qkv, q_gain, k_gain, cos_tab, sin_tab, dqkv, dq_gain, dk_gain, accumulate_dw, workspace, M,
                             T, Hq, Hkv, eps, nullptr, nullptr, stream);
}

extern "C" int sd_swiglu_fwd(const void* gate_up, void* act, int M, int I, void* stream) {
  if (M <= 0 || (I & 7)) return SD_ERR_SHAPE;
  const long n8 = (long)M * I / 8;
  const int nb = (int)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096);
  SdProfScope prof(SD_K_SWIGLU, 6.0 * M * I, ST);
  SD_PROF_LABEL("swiglu_fwd_kernel");
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(nb), dim3(256), 0, ST, (const bf16*)gate_up, (bf16*)act, n8, I);
  SD_CHECK_LAUNCH();
  return 0;
}

extern "C" int sd_swiglu_bwd(const void* dact, const void* gate_up, void* dgate_up, int M, int I, void* stream) {
  if (M <= 0 || (I & 7)) return SD_ERR_SHAPE;
  const long n8 = (long)M * I / 8;
  const int nb = (int)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096);
  SdProfScope prof(SD_K_SWIGLU, 10.0 * M * I, ST);
  SD_PROF_LABEL("swiglu_bwd_kernel");
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(nb), dim3(256), 0, ST, (const bf16*)dact, (const bf16*)gate_up,
                     (bf16*)dgate_up, n8, I);
  SD_CHECK_LAUNCH();
  return 0;
}

extern "C" int sd_embedding_fwd(const int64_t* ids, const void* E, void* x, int M, int H, int V, void* stream) {
  if (M <= 0 || (H & 7)) return SD_ERR_SHAPE;
  SdProfScope prof(SD_K_EMBED, 4.0 * M * H, ST);
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, ST, ids, (const bf16*)E, (bf16*)x, M, H, V);
  SD_CHECK_LAUNCH();
  return 0;
}

extern "C" int sd_embedding_fwd_ssq(const int64_t* ids, const void* E, void* x, float* ssq_out, int M, int H, int V,
                                    void* stream) {
  if (M <= 0 || (H & 7) || !ssq_out) return SD_ERR_SHAPE;
  if ((H % 512) || H / 128 > 16) return SD_ERR_UNSUPPORTED;
  SdProfScope prof(SD_K_EMBED, 4.0 * M * H, ST);
  hipLaunchKernelGGL(embedding_fwd_ssq_kernel, dim3((M + 15) / 16), dim3(256), 0, ST, ids, (const bf16*)E, (bf16*)x, ssq_out, M,
                     H, V);
  SD_CHECK_LAUNCH();
  return 0;
}

extern "C" int sd_rows_scatter(const void* src, const int64_t* rows, void* dst, int n, int M, int H, void* stream) {
  if (n < 0 || M <= 0 || (H & 7)) return SD_ERR_SHAPE;
  SdProfScope prof(SD_K_EMBED, 2.0 * M * H + 4.0 * n * H, ST);
  if (hipMemsetAsync(dst, 0, (size_t)M * H * 2, ST) != hipSuccess) return SD_ERR_WORKSPACE;
  if (n == 0) return 0;
  hipLaunchKernelGGL(rows_scatter_kernel, dim3((n + 3) / 4), dim3(256), 0, ST, (const bf16*)src, rows, (bf16*)dst, n, M, H);
  SD_CHECK_LAUNCH();
  return 0;
}

extern "C" int sd_embedding_bwd(const int64_t* ids, const void* dx, void* dE, int M, int H, int V, float scale,
                                void* stream) {
  if (M <= 0 || (H & 7)) return SD_ERR_SHAPE;
  SdProfScope prof(SD_K_EMBED, 6.0 * M * H, ST);
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3(M), dim3(256), 0, ST, ids, (const bf16*)dx, (bf16*)dE, M, H, V, scale);
  SD_CHECK_LAUNCH();
  return 0;
}
