// Fused AdamW on bf16 parameters with bf16 moments, and the global gradient norm used for clipping.
// Replaces the HF Trainer default optimizer step on the bf16 student (train.py:174 loads the model in
// bf16, train.py:331-354 sets no `optim`) and Trainer's clip_grad_norm_(1.0).
// HBM-bound: 4 streams read + 3 written per parameter, 16 bytes per lane per access.
#include "sd_common.cuh"
#include "../../include/sd_hip.h"
#include "sd_prof.h"

namespace {

// Per-workgroup partial sums of sd_sumsq_bf16, combined by a second launch in a fixed order: the norm must come out
// bit-identical on every data-parallel rank (it scales the update, and ranks that round it differently drift apart --
// seen as different parameter checksums on two ranks when the partials were combined with atomicAdd).  The partials
// live in the CALLER's workspace (SD_SUMSQ_PARTIALS floats), so calls on different streams never share state.
constexpr int kSumsqPartials = SD_SUMSQ_PARTIALS;

__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partials, int nb, float* out) {
  __shared__ float sc[32];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += partials[i];
  s = block_sum<256>(s, sc);
  if (threadIdx.x == 0) out[0] += s;
}

__global__ __launch_bounds__(256) void sumsq_kernel(const bf16* __restrict__ x, long n8, long n, float* partials) {
  __shared__ float sc[32];
  float s = 0.f;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n8; q += (long)gridDim.x * 256) {
    bf16x8 v = *(const bf16x8*)(x + q * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s += f * f; }
  }
  if (blockIdx.x == 0)
    for (long i = n8 * 8 + threadIdx.x; i < n; i += 256) { const float f = (float)x[i]; s += f * f; }
  s = block_sum<256>(s, sc);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

SD_DEV void adam1(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps, float wd, float bc1,
                  float rbc2) {
  p *= (1.f - lr * wd);
  m = b1 * m + (1.f - b1) * g;
  v = b2 * v + (1.f - b2) * g * g;
  const float denom = sqrtf(v) * rbc2 + eps;
  p -= (lr / bc1) * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_kernel(bf16* p, const bf16* __restrict__ g, bf16* m, bf16* v, long n8,
                                                    long n, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                    float rbc2, const float* __restrict__ sumsq, float max_norm) {
  float clip = 1.f;
  if (sumsq && max_norm > 0.f) {
    const float nrm = sqrtf(sumsq[0]);
    clip = fminf(1.f, max_norm / (nrm + 1e-6f));
  }
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n8; q += (long)gridDim.x * 256) {
    bf16x8 pv = *(const bf16x8*)(p + q * 8), gv = *(const bf16x8*)(g + q * 8);
    bf16x8 mv = *(const bf16x8*)(m + q * 8), vv = *(const bf16x8*)(v + q * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float pf = (float)pv[e], mf = (float)mv[e], vf = (float)vv[e];
      adam1(pf, (float)gv[e] * clip, mf, vf, lr, b1, b2, eps, wd, bc1, rbc2);
      pv[e] = (bf16)pf; mv[e] = (bf16)mf; vv[e] = (bf16)vf;
    }
    *(bf16x8*)(p + q * 8) = pv; *(bf16x8*)(m + q * 8) = mv; *(bf16x8*)(v + q * 8) = vv;
  }
  if (blockIdx.x == 0)
    for (long i = n8 * 8 + threadIdx.x; i < n; i += 256) {
      float pf = (float)p[i], mf = (float)m[i], vf = (float)v[i];
      adam1(pf, (float)g[i] * clip, mf, vf, lr, b1, b2, eps, wd, bc1, rbc2);
      p[i] = (bf16)pf; m[i] = (bf16)mf; v[i] = (bf16)vf;
    }
}

}  // namespace

extern "C" int sd_sumsq_bf16(const void* x, int64_t n, float* out_accum, float* partials, void* stream) {
  if (n <= 0) return SD_ERR_SHAPE;
  if (!partials) return SD_ERR_WORKSPACE;
  if (((uintptr_t)x & 15) || ((uintptr_t)partials & 3)) return SD_ERR_ALIGN;
  const long n8 = n / 8;
  const int nb = (int)((n8 + 255) / 256 < kSumsqPartials ? (n8 + 255) / 256 : kSumsqPartials);
  const int grid = nb < 1 ? 1 : nb;
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, n8, (long)n, partials);
  SD_CHECK_LAUNCH();
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partials, grid,
                     out_accum);
  SD_CHECK_LAUNCH();
  return 0;
}

extern "C" int sd_adamw_bf16(void* param, const void* grad, void* exp_avg, void* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int step, const float* grad_sumsq,
                             float max_grad_norm, void* stream) {
  if (n <= 0 || step <= 0) return SD_ERR_SHAPE;
  if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return SD_ERR_ALIGN;
  const long n8 = n / 8;
  const int nb = (int)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096);
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float rbc2 = 1.f / sqrtf(1.f - powf(beta2, (float)step));
  SdProfScope prof(SD_K_OPTIM, 14.0 * n, (hipStream_t)stream);
  hipLaunchKernelGGL(adamw_kernel, dim3(nb < 1 ? 1 : nb), dim3(256), 0, (hipStream_t)stream, (bf16*)param,
                     (const bf16*)grad, (bf16*)exp_avg, (bf16*)exp_avg_sq, n8, (long)n, lr, beta1, beta2, eps,
                     weight_decay, bc1, rbc2, grad_sumsq, max_grad_norm);
  SD_CHECK_LAUNCH();
  return 0;
}
