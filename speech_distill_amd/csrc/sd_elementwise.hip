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
#include "sd_common.cuh"
#include "../../include/sd_hip.h"

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
// One wave per row; the row is read twice (second read is L1/L2 resident).
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                          bf16* __restrict__ y, float* __restrict__ rstd_out, int M,
                                                          int H, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = lane_id();
  const bf16* xr = x + (long)row * H;
  float ss = 0.f;
  for (int c = lane * 8; c < H; c += 512) {
    float f[8];
    unpack8(*(const bf16x8*)(xr + c), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += f[e] * f[e];
  }
  ss = wave_sum(ss);
  const float rstd = rsqrtf(ss / (float)H + eps);
  if (lane == 0 && rstd_out) rstd_out[row] = rstd;
  bf16* yr = y + (long)row * H;
  for (int c = lane * 8; c < H; c += 512) {
    float f[8], g[8];
    unpack8(*(const bf16x8*)(xr + c), f);
    unpack8(*(const bf16x8*)(w + c), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = g[e] * (float)(bf16)(f[e] * rstd);  // HF: weight * hidden.to(bf16)
    *(bf16x8*)(yr + c) = pack8(f);
  }
}

// dx = rstd * (g - xhat * mean(g*xhat)),  g = dy*w, xhat = x*rstd;  dx (+)= dres.
// dw partials: block b owns rows [b*rpb, (b+1)*rpb); thread t owns columns 8t.. (H <= 2048*... looped).
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                          const bf16* __restrict__ w, const float* __restrict__ rstd,
                                                          const bf16* dres, bf16* dx, float* __restrict__ dw_part,
                                                          int M, int H, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float dw_s[];  // [H]
  for (int c = threadIdx.x; c < H; c += 256) dw_s[c] = 0.f;
  __syncthreads();
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  for (int row = r0 + wv; row < r1; row += 4) {
    const bf16* xr = x + (long)row * H;
    const bf16* dyr = dy + (long)row * H;
    const float rs = rstd[row];
    float dot = 0.f;
    for (int c = lane * 8; c < H; c += 512) {
      float xf[8], df[8], wf[8];
      unpack8(*(const bf16x8*)(xr + c), xf);
      unpack8(*(const bf16x8*)(dyr + c), df);
      unpack8(*(const bf16x8*)(w + c), wf);
#pragma unroll
      for (int e = 0; e < 8; ++e) dot += df[e] * wf[e] * xf[e] * rs;
    }
    dot = wave_sum(dot) / (float)H;
    for (int c = lane * 8; c < H; c += 512) {
      float xf[8], df[8], wf[8], o[8];
      unpack8(*(const bf16x8*)(xr + c), xf);
      unpack8(*(const bf16x8*)(dyr + c), df);
      unpack8(*(const bf16x8*)(w + c), wf);
      if (dres) unpack8(*(const bf16x8*)(dres + (long)row * H + c), o);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = xf[e] * rs;
        const float v = rs * (df[e] * wf[e] - xh * dot);
        o[e] = dres ? o[e] + v : v;
        atomicAdd(&dw_s[c + e], df[e] * xh);  // LDS atomic: 4 waves share the column sums
      }
      *(bf16x8*)(dx + (long)row * H + c) = pack8(o);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 256) dw_part[(long)blockIdx.x * H + c] = dw_s[c];
}

// out[c] (bf16) = (accumulate ? out[c] : 0) + sum_b part[b][c]
__global__ void colsum_reduce_kernel(const float* __restrict__ part, bf16* out, int nb, int H, int stride,
                                     int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= H) return;
  float s = 0.f;
  for (int b = 0; b < nb; ++b) s += part[(long)b * stride + c];
  if (accumulate) s += (float)out[c];
  out[c] = (bf16)s;
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
  ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64); ss += __shfl_xor(ss, 8, 64);
  const float rs = rsqrtf(ss * (1.f / 128.f) + eps);
  unpack8(*(const bf16x8*)((hh < Hq ? qw : kw) + j * 8), g);
  unpack8(*(const bf16x8*)(cosb + (long)t * 128 + j * 8), cs);
  unpack8(*(const bf16x8*)(sinb + (long)t * 128 + j * 8), sn);
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float n = (float)(bf16)(g[e] * (float)(bf16)(f[e] * rs));  // normed value as HF holds it (bf16)
    const float p = __shfl_xor(n, 8, 64);
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
  __shared__ float dw_s[256];
  dw_s[threadIdx.x] = 0.f;
  __syncthreads();
  const int lane = lane_id();
  const int sub = lane >> 4, j = lane & 15;
  const int nh = Hq + Hkv;
  const long total = (long)M * nh;
  const long i0 = (long)blockIdx.x * items_per_block;
  float accw[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // this lane's gain-gradient partial for q (hh<Hq) ...
  float acck[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // ... and for k
  for (long base = i0 + (threadIdx.x >> 6) * 4; base < i0 + items_per_block; base += 16) {
    const long idx = base + sub;
    const bool ok = idx < total && idx < i0 + items_per_block;
    const long id = idx < total ? idx : total - 1;
    const int m = (int)(id / nh), hh = (int)(id % nh);
    const int t = m % T;
    float f[8], g[8], cs[8], sn[8], dy[8];
    unpack8(*(const bf16x8*)(qkv + (long)m * (Hq + 2 * Hkv) * 128 + hh * 128 + j * 8), f);
    unpack8(*(const bf16x8*)(dout + (long)m * nh * 128 + hh * 128 + j * 8), dy);
    unpack8(*(const bf16x8*)((hh < Hq ? qw : kw) + j * 8), g);
    unpack8(*(const bf16x8*)(cosb + (long)t * 128 + j * 8), cs);
    unpack8(*(const bf16x8*)(sinb + (long)t * 128 + j * 8), sn);
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += f[e] * f[e];
    ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64); ss += __shfl_xor(ss, 8, 64);
    const float rs = rsqrtf(ss * (1.f / 128.f) + eps);
    // RoPE^T:  dn[i] = dy[i] cos[i] + (i<64 ?  dy[i+64] sin[i+64] : -dy[i-64] sin[i-64])
    float dn[8], dot = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float z = dy[e] * sn[e];
      const float pz = __shfl_xor(z, 8, 64);
      dn[e] = dy[e] * cs[e] + ((j < 8) ? pz : -pz);
      dot += dn[e] * g[e] * f[e] * rs;
    }
    dot += __shfl_xor(dot, 1, 64); dot += __shfl_xor(dot, 2, 64); dot += __shfl_xor(dot, 4, 64); dot += __shfl_xor(dot, 8, 64);
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
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    atomicAdd(&dw_s[j * 8 + e], accw[e]);
    atomicAdd(&dw_s[128 + j * 8 + e], acck[e]);
  }
  __syncthreads();
  dw_part[(long)blockIdx.x * 256 + threadIdx.x] = dw_s[threadIdx.x];
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

// Deterministic scatter-add: the block of the FIRST token carrying an id sums every token row with
// that id (fixed order) and adds the total to dE[id]; other blocks exit.  No atomics, no sort.
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int64_t* __restrict__ ids, const bf16* __restrict__ dx,
                                                            bf16* dE, int M, int H, int V) {
  __shared__ int first_flag;
  const int m = blockIdx.x;
  const long id = ids[m];
  if (id < 0 || id >= V) return;
  if (threadIdx.x == 0) first_flag = 1;
  __syncthreads();
  for (int i = threadIdx.x; i < m; i += 256)
    if (ids[i] == id) first_flag = 0;
  __syncthreads();
  if (!first_flag) return;
  for (int c = threadIdx.x * 8; c < H; c += 2048) {
    float acc[8], f[8];
    unpack8(*(const bf16x8*)(dE + id * H + c), acc);
    for (int i = m; i < M; ++i) {
      if (ids[i] != id) continue;
      unpack8(*(const bf16x8*)(dx + (long)i * H + c), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += f[e];
    }
    *(bf16x8*)(dE + id * H + c) = pack8(acc);
  }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int sd_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int M, int H, float eps, void* stream) {
  if (M <= 0 || (H & 7)) return SD_ERR_SHAPE;
  hipLaunchKernelGGL(rmsnorm_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, ST, (const bf16*)x, (const bf16*)w, (bf16*)y,
                     rstd, M, H, eps);
  SD_CHECK_LAUNCH();
  return 0;
}

extern "C" int64_t sd_rmsnorm_bwd_workspace_bytes(int M, int H) {
  const int nb = (M + 7) / 8 < 256 ? (M + 7) / 8 : 256;
  return (int64_t)nb * H * 4;
}

extern "C" int sd_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx,
                              void* dw, int accumulate_dw, void* workspace, int M, int H, void* stream) {
  if (M <= 0 || (H & 7) || H > 16384) return SD_ERR_SHAPE;
  int nb = (M + 7) / 8 < 256 ? (M + 7) / 8 : 256;
  const int rpb = (M + nb - 1) / nb;
  nb = (M + rpb - 1) / rpb;
  hipLaunchKernelGGL(rmsnorm_bwd_kernel, dim3(nb), dim3(256), H * 4, ST, (const bf16*)dy, (const bf16*)x, (const bf16*)w,
                     rstd, (const bf16*)dres, (bf16*)dx, (float*)workspace, M, H, rpb);
  SD_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((H + 255) / 256), dim3(256), 0, ST, (const float*)workspace, (bf16*)dw,
                     nb, H, H, accumulate_dw);
  SD_CHECK_LAUNCH();
  return 0;
}

extern "C" int sd_qknorm_rope_fwd(const void* qkv, const void* q_gain, const void* k_gain, const void* cos_tab,
                                  const void* sin_tab, void* qk_out, int M, int T, int Hq, int Hkv, float eps,
                                  void* stream) {
  if (M <= 0 || T <= 0 || (M % T)) return SD_ERR_SHAPE;
  const long items = (long)M * (Hq + Hkv);
  hipLaunchKernelGGL(qknorm_rope_fwd_kernel, dim3((unsigned)((items + 15) / 16)), dim3(256), 0, ST, (const bf16*)qkv,
                     (const bf16*)q_gain, (const bf16*)k_gain, (const bf16*)cos_tab, (const bf16*)sin_tab, (bf16*)qk_out,
                     M, T, Hq, Hkv, eps);
  SD_CHECK_LAUNCH();
  return 0;
}

static inline int qk_bwd_blocks(long items, int* ipb) {
  long per = (items + 511) / 512;
  per = (per + 15) / 16 * 16;
  *ipb = (int)per;
  return (int)((items + per - 1) / per);
}

extern "C" int64_t sd_qknorm_rope_bwd_workspace_bytes(int M, int Hq, int Hkv) {
  int ipb;
  return (int64_t)qk_bwd_blocks((long)M * (Hq + Hkv), &ipb) * 256 * 4;
}

extern "C" int sd_qknorm_rope_bwd(const void* dqk, const void* qkv, const void* q_gain, const void* k_gain,
                                  const void* cos_tab, const void* sin_tab, void* dqkv, void* dq_gain, void* dk_gain,
                                  int accumulate_dw, void* workspace, int M, int T, int Hq, int Hkv, float eps,
                                  void* stream) {
  if (M <= 0 || T <= 0 || (M % T)) return SD_ERR_SHAPE;
  int ipb;
  const int nb = qk_bwd_blocks((long)M * (Hq + Hkv), &ipb);
  hipLaunchKernelGGL(qknorm_rope_bwd_kernel, dim3(nb), dim3(256), 0, ST, (const bf16*)dqk, (const bf16*)qkv,
                     (const bf16*)q_gain, (const bf16*)k_gain, (const bf16*)cos_tab, (const bf16*)sin_tab, (bf16*)dqkv,
                     (float*)workspace, M, T, Hq, Hkv, eps, ipb);
  SD_CHECK_LAUNCH();
  // partial layout [nb][256]: columns 0..127 -> q gain, 128..255 -> k gain
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3(1), dim3(128), 0, ST, (const float*)workspace, (bf16*)dq_gain, nb, 128,
                     256, accumulate_dw);
  SD_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3(1), dim3(128), 0, ST, (const float*)workspace + 128, (bf16*)dk_gain,
                     nb, 128, 256, accumulate_dw);
  SD_CHECK_LAUNCH();
  return 0;
}

extern "C" int sd_swiglu_fwd(const void* gate_up, void* act, int M, int I, void* stream) {
  if (M <= 0 || (I & 7)) return SD_ERR_SHAPE;
  const long n8 = (long)M * I / 8;
  const int nb = (int)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096);
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(nb), dim3(256), 0, ST, (const bf16*)gate_up, (bf16*)act, n8, I);
  SD_CHECK_LAUNCH();
  return 0;
}

extern "C" int sd_swiglu_bwd(const void* dact, const void* gate_up, void* dgate_up, int M, int I, void* stream) {
  if (M <= 0 || (I & 7)) return SD_ERR_SHAPE;
  const long n8 = (long)M * I / 8;
  const int nb = (int)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096);
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(nb), dim3(256), 0, ST, (const bf16*)dact, (const bf16*)gate_up,
                     (bf16*)dgate_up, n8, I);
  SD_CHECK_LAUNCH();
  return 0;
}

extern "C" int sd_embedding_fwd(const int64_t* ids, const void* E, void* x, int M, int H, int V, void* stream) {
  if (M <= 0 || (H & 7)) return SD_ERR_SHAPE;
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, ST, ids, (const bf16*)E, (bf16*)x, M, H, V);
  SD_CHECK_LAUNCH();
  return 0;
}

extern "C" int sd_embedding_bwd(const int64_t* ids, const void* dx, void* dE, int M, int H, int V, void* stream) {
  if (M <= 0 || (H & 7)) return SD_ERR_SHAPE;
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3(M), dim3(256), 0, ST, ids, (const bf16*)dx, (bf16*)dE, M, H, V);
  SD_CHECK_LAUNCH();
  return 0;
}
