// LoRA student (train.py:180-202: peft LoRA on the seven projections of every decoder layer), done on the MERGED weight.
//
// The decoder GEMMs never see the adapter: once per optimizer step every target weight is rebuilt as
//     W_eff = W_res + (s B) A          (sd_lora_merge;  A [r,in], B [out,r], s = alpha / sqrt(r))
// the backward writes the ordinary full weight gradient dW into the flat gradient buffer, and before the optimizer runs
//     dA = (s B)^T dW,   dB = dW (s A)^T          (sd_lora_project)
// which is the chain rule through the merge.  All three are HBM-bound passes over the 0.44 G projection weights with a
// rank-r (r = 32: ONE 16x16x32 MFMA k-step) product riding along, so they are table-driven single launches over all
// 7 L targets rather than 3 x 7 L skinny GEMMs:
//     merge     reads W_res, writes W_eff              4 B / weight
//     project   reads dW twice (dB: row blocks over all columns; dA: column blocks over all rows -- each output is
//               owned by ONE workgroup, so no atomics: the result must be bit-identical on every data-parallel rank)
// A and B are fp32 masters (peft keeps adapter weights in fp32 next to a bf16 base model); sd_adamw_f32_shadow updates
// them and emits the bf16 operands the kernels above read: shadow = bf16(p) and scaled shadow = bf16(s p).
//
// MFMA operand layouts (v_mfma_f32_16x16x32_bf16, D = X Y):  X: lane l holds X[l%16][8(l/16) .. +8],
// Y: lane l holds Y[8(l/16) .. +8][l%16],  D: lane l holds D[4(l/16) + i][l%16], i = 0..3.
#include <type_traits>
#include "sd_common.cuh"
#include "../../include/sd_hip.h"
#include "sd_prof.h"

namespace {

struct PlanHeader { int n, r_pad, merge_items, db_items, da_items, pad[3]; };
struct PlanEntry {
  const bf16* w_res; bf16* w_out; const bf16* w_grad;
  const bf16* a_sh; const bf16* a_scaled; const bf16* b_scaled;
  bf16* d_a; bf16* d_b;
  int out_f, in_f, merge_base, db_base, da_base, col_blocks, pad[2];
};
static_assert(sizeof(PlanHeader) == 32 && sizeof(PlanEntry) == 96, "plan layout");

constexpr int kMergeRows = 256;  // rows one merge item walks (16 MFMA row tiles) with its A operand held in registers
constexpr int kColBlock = 128;   // columns per merge / dA item
constexpr int kDbRows = 64;      // rows per dB item (one workgroup)

template <int WHICH> SD_DEV int plan_find(const PlanEntry* e, int n, int item) {
  int lo = 0, hi = n - 1;  // last entry whose base <= item
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    const int b = WHICH == 0 ? e[mid].merge_base : (WHICH == 1 ? e[mid].db_base : e[mid].da_base);
    if (b <= item) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// ---------------------------------------------------------------------------------------------------------------- merge
// One wave per item = (target, 128-column block, 256-row chunk).  Column (l%16) of MFMA block j is the weight column
// 8 (l%16) + j of the block, so a lane's 8 accumulators of one output row are 8 CONSECUTIVE columns: W_res is read and
// W_eff written as 16-byte vectors, 256 contiguous bytes per row per wave.
template <int KS>
__global__ __launch_bounds__(256) void lora_merge_kernel(const PlanHeader* __restrict__ plan) {
  const PlanEntry* ents = (const PlanEntry*)(plan + 1);
  const int item = blockIdx.x * 4 + wave_id_uniform();
  if (item >= plan->merge_items) return;
  const PlanEntry& E = ents[plan_find<0>(ents, plan->n, item)];
  const int local = item - E.merge_base;
  const int cb = local % E.col_blocks, rc = local / E.col_blocks;
  const int in_f = E.in_f, r_pad = KS * 32;
  const int l = lane_id(), lr = l & 15, lg = l >> 4;
  const int col0 = cb * kColBlock + lr * 8;
  // Y operands: y[ks][j][e] = A[ks*32 + 8 lg + e][col0 + j]  (8 vector loads per k-step, transposed in registers)
  bf16x8 y[KS][8];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bf16x8 rows[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) rows[e] = *(const bf16x8*)(E.a_sh + (long)(ks * 32 + lg * 8 + e) * in_f + col0);
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) y[ks][j][e] = rows[e][j];
  }
  const int row_begin = rc * kMergeRows;
  const int row_end = min(E.out_f, row_begin + kMergeRows);
  for (int r0 = row_begin; r0 < row_end; r0 += 16) {
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 x = *(const bf16x8*)(E.b_scaled + (long)(r0 + lr) * r_pad + ks * 32 + lg * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = mfma16(x, y[ks][j], acc[j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long off = (long)(r0 + lg * 4 + i) * in_f + col0;
      bf16x8 w = *(const bf16x8*)(E.w_res + off);
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = (bf16)((float)w[j] + acc[j][i]);
      *(bf16x8*)(E.w_out + off) = w;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------- dB
// dB[out, r_pad] = dW[out, in] (sA)^T: both operands are K-contiguous.  One workgroup per 64 rows walks K 128 columns at
// a time: the [64 x 128] tile of dW and the [r_pad x 128] tile of sA are fetched with row-contiguous 16-byte loads
// (a wave instruction = 4 rows x 256 B; fragment-layout loads straight from global memory -- 16 rows x 64 B per
// instruction -- measured 2.3-3.1 TB/s), double-buffered through LDS, and read back as ds_read_b128 MFMA fragments.
// Wave w owns rows 16 w .. 16 w + 16 for all of K: every output element has one owner, no reduction.
constexpr int kDbTiles = kDbRows / 16;
constexpr int kDbK = 128, kDbS = kDbK + 8;  // LDS row stride (elements): 272 B, conflict-free for b128 reads and writes
template <int NB>
__global__ __launch_bounds__(256) void lora_db_kernel(const PlanHeader* __restrict__ plan) {
  constexpr int R = NB * 16;
  constexpr int YV = R * kDbK / 8 / 256;  // 16-byte vectors of the sA tile per thread
  static_assert(kDbTiles == 4 && YV >= 1, "one row tile per wave");
  __shared__ __attribute__((aligned(16))) bf16 lx[2][kDbRows * kDbS];
  __shared__ __attribute__((aligned(16))) bf16 ly[2][R * kDbS];
  const PlanEntry* ents = (const PlanEntry*)(plan + 1);
  const int item = blockIdx.x;
  const PlanEntry& E = ents[plan_find<1>(ents, plan->n, item)];
  const int row0 = (item - E.db_base) * kDbRows;
  const int in_f = E.in_f;
  const int tid = threadIdx.x, l = tid & 63, lr = l & 15, lg = l >> 4, wv = wave_id_uniform();
  const int rows_here = min(kDbRows, E.out_f - row0);  // out_f % 32 == 0: a short last block has 32 rows
  const int srow = tid >> 4, scol = (tid & 15) * 8;
  u32x4 xr[4], yr[YV];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = srow + 16 * q;
      xr[q] = *(const u32x4*)(E.w_grad + (long)(row0 + (r < rows_here ? r : 0)) * in_f + k0 + scol);
    }
#pragma unroll
    for (int q = 0; q < YV; ++q) yr[q] = *(const u32x4*)(E.a_scaled + (long)(srow + 16 * q) * in_f + k0 + scol);
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) *(u32x4*)&lx[buf][(srow + 16 * q) * kDbS + scol] = xr[q];
#pragma unroll
    for (int q = 0; q < YV; ++q) *(u32x4*)&ly[buf][(srow + 16 * q) * kDbS + scol] = yr[q];
  };
  f32x4 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  fetch(0);
  stash(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < in_f; k0 += kDbK, buf ^= 1) {   // in_f % 128 == 0
    const bool more = k0 + kDbK < in_f;
    if (more) fetch(k0 + kDbK);
#pragma unroll
    for (int ks = 0; ks < kDbK / 32; ++ks) {
      const bf16x8 x = *(const bf16x8*)&lx[buf][(wv * 16 + lr) * kDbS + ks * 32 + lg * 8];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const bf16x8 y = *(const bf16x8*)&ly[buf][(nb * 16 + lr) * kDbS + ks * 32 + lg * 8];
        acc[nb] = mfma16(x, y, acc[nb]);
      }
    }
    if (more) stash(buf ^ 1);
    __syncthreads();
  }
  if (wv * 16 < rows_here) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        E.d_b[(long)(row0 + wv * 16 + lg * 4 + i) * R + nb * 16 + lr] = (bf16)acc[nb][i];
  }
}

// ------------------------------------------------------------------------------------------------------------------- dA
// dA[r_pad, in] = (sB)^T dW: the contraction runs over ROWS of both operands, so both go through LDS and are read back
// column-wise.  One workgroup per (target, 128-column block) walks all rows 32 at a time (double-buffered); wave w
// owns columns 32 w .. 32 w + 32 of the block and every row block of dA.
constexpr int kDaSW = kColBlock + 2;  // LDS row stride (elements) of the dW tile: 8 rows apart = 8 banks apart
template <int MB>
__global__ __launch_bounds__(256) void lora_da_kernel(const PlanHeader* __restrict__ plan) {
  constexpr int R = MB * 16, SB = R + 2;
  constexpr int BVEC = 32 * R / 8;  // 16-byte vectors in one (sB) tile
  __shared__ uint32_t lds_w[2][32 * kDaSW / 2];
  __shared__ uint32_t lds_b[2][32 * SB / 2];
  const PlanEntry* ents = (const PlanEntry*)(plan + 1);
  const int item = blockIdx.x;
  const PlanEntry& E = ents[plan_find<2>(ents, plan->n, item)];
  const int cb = item - E.da_base;
  const int in_f = E.in_f, out_f = E.out_f;
  const int tid = threadIdx.x, l = tid & 63, lr = l & 15, lg = l >> 4, wv = wave_id_uniform();
  // staging roles: dW tile rows tid/16 and tid/16 + 16, 8 columns at 8 (tid%16)
  const int srow = tid >> 4, scol = (tid & 15) * 8;
  const bf16* gsrc = E.w_grad + (long)srow * in_f + cb * kColBlock + scol;
  u32x4 w0, w1, bv[(BVEC + 255) / 256];
  auto fetch = [&](int k0) {
    w0 = *(const u32x4*)(gsrc + (long)k0 * in_f);
    w1 = *(const u32x4*)(gsrc + (long)(k0 + 16) * in_f);
#pragma unroll
    for (int q = 0; q < (BVEC + 255) / 256; ++q) {
      const int v = tid + q * 256;
      if (v < BVEC) bv[q] = *(const u32x4*)(E.b_scaled + (long)k0 * R + (long)v * 8);
    }
  };
  auto stash = [&](int buf) {
    uint32_t* d0 = &lds_w[buf][(srow * kDaSW + scol) / 2];
    uint32_t* d1 = &lds_w[buf][((srow + 16) * kDaSW + scol) / 2];
#pragma unroll
    for (int c = 0; c < 4; ++c) { d0[c] = w0[c]; d1[c] = w1[c]; }
#pragma unroll
    for (int q = 0; q < (BVEC + 255) / 256; ++q) {
      const int v = tid + q * 256;
      if (v < BVEC) {
        const int row = (v * 8) / R, col = (v * 8) % R;
        uint32_t* d = &lds_b[buf][(row * SB + col) / 2];
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = bv[q][c];
      }
    }
  };
  f32x4 acc[MB][2];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb][0] = acc[mb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  fetch(0);
  stash(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < out_f; k0 += 32, buf ^= 1) {
    const bool more = k0 + 32 < out_f;
    if (more) fetch(k0 + 32);
    const bf16* tw = (const bf16*)lds_w[buf];
    const bf16* tb = (const bf16*)lds_b[buf];
    bf16x8 y[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 8; ++e) y[nb][e] = tw[(lg * 8 + e) * kDaSW + wv * 32 + nb * 16 + lr];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      bf16x8 x;
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = tb[(lg * 8 + e) * SB + mb * 16 + lr];
      acc[mb][0] = mfma16(x, y[0], acc[mb][0]);
      acc[mb][1] = mfma16(x, y[1], acc[mb][1]);
    }
    if (more) stash(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        E.d_a[(long)(mb * 16 + lg * 4 + i) * in_f + cb * kColBlock + wv * 32 + nb * 16 + lr] = (bf16)acc[mb][nb][i];
}

// ---------------------------------------------------------------------------------------------------- fp32-master AdamW
__global__ __launch_bounds__(256) void adamw_f32_shadow_kernel(float* p, const bf16* __restrict__ g, float* m, float* v,
                                                               bf16* shadow, bf16* shadow_scaled, float scale, long n4,
                                                               float lr, float b1, float b2, float eps, float wd, float bc1,
                                                               float rbc2, const float* __restrict__ sumsq, float max_norm) {
  float clip = 1.f;
  if (sumsq && max_norm > 0.f) clip = fminf(1.f, max_norm / (sqrtf(sumsq[0]) + 1e-6f));
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long)gridDim.x * 256) {
    f32x4 pv = *(const f32x4*)(p + q * 4), mv = *(const f32x4*)(m + q * 4), vv = *(const f32x4*)(v + q * 4);
    const bf16x4 gv = *(const bf16x4*)(g + q * 4);
    bf16x4 s1, s2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pf = pv[e], mf = mv[e], vf = vv[e];
      const float gf = (float)gv[e] * clip;
      pf *= (1.f - lr * wd);
      mf = b1 * mf + (1.f - b1) * gf;
      vf = b2 * vf + (1.f - b2) * gf * gf;
      pf -= (lr / bc1) * (mf / (sqrtf(vf) * rbc2 + eps));
      pv[e] = pf; mv[e] = mf; vv[e] = vf;
      s1[e] = (bf16)pf; s2[e] = (bf16)(pf * scale);
    }
    *(f32x4*)(p + q * 4) = pv; *(f32x4*)(m + q * 4) = mv; *(f32x4*)(v + q * 4) = vv;
    *(bf16x4*)(shadow + q * 4) = s1; *(bf16x4*)(shadow_scaled + q * 4) = s2;
  }
}

template <typename F> int pick_r(int r_pad, F&& f) {
  switch (r_pad) {
    case 32: return f(std::integral_constant<int, 1>{});
    case 64: return f(std::integral_constant<int, 2>{});
    case 128: return f(std::integral_constant<int, 4>{});
    default: return SD_ERR_UNSUPPORTED;
  }
}

}  // namespace

extern "C" int64_t sd_lora_plan_bytes(int n_targets) {
  return n_targets > 0 ? (int64_t)sizeof(PlanHeader) + (int64_t)n_targets * sizeof(PlanEntry) : 0;
}

extern "C" int sd_lora_plan_build(const SdLoraTarget* t, int n, int r_pad, void* plan_host, int64_t plan_bytes) {
  if (!t || n <= 0 || !plan_host) return SD_ERR_SHAPE;
  if (plan_bytes < sd_lora_plan_bytes(n)) return SD_ERR_WORKSPACE;
  if (r_pad != 32 && r_pad != 64 && r_pad != 128) return SD_ERR_UNSUPPORTED;
  PlanHeader* h = (PlanHeader*)plan_host;
  PlanEntry* e = (PlanEntry*)(h + 1);
  long merge = 0, db = 0, da = 0;
  for (int i = 0; i < n; ++i) {
    const SdLoraTarget& s = t[i];
    if (s.out_features <= 0 || s.in_features <= 0) return SD_ERR_SHAPE;
    if ((s.in_features % kColBlock) || (s.out_features % 32)) return SD_ERR_UNSUPPORTED;
    if (((uintptr_t)s.w_res | (uintptr_t)s.w_out | (uintptr_t)s.w_grad | (uintptr_t)s.a_shadow | (uintptr_t)s.a_scaled |
         (uintptr_t)s.b_scaled | (uintptr_t)s.d_a | (uintptr_t)s.d_b) & 15) return SD_ERR_ALIGN;
    if (!s.w_res || !s.w_out || !s.w_grad || !s.a_shadow || !s.a_scaled || !s.b_scaled || !s.d_a || !s.d_b) return SD_ERR_SHAPE;
    PlanEntry& p = e[i];
    p.w_res = (const bf16*)s.w_res; p.w_out = (bf16*)s.w_out; p.w_grad = (const bf16*)s.w_grad;
    p.a_sh = (const bf16*)s.a_shadow; p.a_scaled = (const bf16*)s.a_scaled; p.b_scaled = (const bf16*)s.b_scaled;
    p.d_a = (bf16*)s.d_a; p.d_b = (bf16*)s.d_b;
    p.out_f = s.out_features; p.in_f = s.in_features;
    p.col_blocks = s.in_features / kColBlock;
    p.merge_base = (int)merge; p.db_base = (int)db; p.da_base = (int)da;
    p.pad[0] = p.pad[1] = 0;
    merge += (long)p.col_blocks * ((s.out_features + kMergeRows - 1) / kMergeRows);
    db += (s.out_features + kDbRows - 1) / kDbRows;
    da += p.col_blocks;
    if (merge > 0x3fffffffL || db > 0x3fffffffL) return SD_ERR_UNSUPPORTED;
  }
  h->n = n; h->r_pad = r_pad; h->merge_items = (int)merge; h->db_items = (int)db; h->da_items = (int)da;
  h->pad[0] = h->pad[1] = h->pad[2] = 0;
  return 0;
}

extern "C" int sd_lora_merge(const void* plan_dev, const void* plan_host, void* stream) {
  if (!plan_dev || !plan_host) return SD_ERR_SHAPE;
  const PlanHeader* h = (const PlanHeader*)plan_host;
  if (h->n <= 0 || h->merge_items <= 0) return SD_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  SdProfScope prof(SD_K_MISC, 0.0, st);
  SD_PROF_LABEL("lora_merge_kernel<%d>", h->r_pad / 32);
  const int rc = pick_r(h->r_pad, [&](auto ks) {
    hipLaunchKernelGGL(lora_merge_kernel<decltype(ks)::value>, dim3((h->merge_items + 3) / 4), dim3(256), 0, st,
                       (const PlanHeader*)plan_dev);
    return 0;
  });
  if (rc) return rc;
  SD_CHECK_LAUNCH();
  return 0;
}

extern "C" int sd_lora_project(const void* plan_dev, const void* plan_host, void* stream) {
  if (!plan_dev || !plan_host) return SD_ERR_SHAPE;
  const PlanHeader* h = (const PlanHeader*)plan_host;
  if (h->n <= 0 || h->db_items <= 0 || h->da_items <= 0) return SD_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  {
    SdProfScope prof(SD_K_MISC, 0.0, st);
    SD_PROF_LABEL("lora_db_kernel<%d>", h->r_pad / 16);
    const int rc = pick_r(h->r_pad, [&](auto ks) {
      hipLaunchKernelGGL(lora_db_kernel<decltype(ks)::value * 2>, dim3(h->db_items), dim3(256), 0, st,
                         (const PlanHeader*)plan_dev);
      return 0;
    });
    if (rc) return rc;
    SD_CHECK_LAUNCH();
  }
  {
    SdProfScope prof(SD_K_MISC, 0.0, st);
    SD_PROF_LABEL("lora_da_kernel<%d>", h->r_pad / 16);
    const int rc = pick_r(h->r_pad, [&](auto ks) {
      hipLaunchKernelGGL(lora_da_kernel<decltype(ks)::value * 2>, dim3(h->da_items), dim3(256), 0, st,
                         (const PlanHeader*)plan_dev);
      return 0;
    });
    if (rc) return rc;
    SD_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int sd_adamw_f32_shadow(float* param, const void* grad, float* exp_avg, float* exp_avg_sq, void* shadow,
                                   void* shadow_scaled, float scale, int64_t n, float lr, float beta1, float beta2,
                                   float eps, float weight_decay, int step, const float* grad_sumsq, float max_grad_norm,
                                   void* stream) {
  if (n <= 0 || step <= 0 || (n & 3)) return SD_ERR_SHAPE;
  if (((uintptr_t)param | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return SD_ERR_ALIGN;
  if (((uintptr_t)grad | (uintptr_t)shadow | (uintptr_t)shadow_scaled) & 7) return SD_ERR_ALIGN;
  const long n4 = n / 4;
  const int nb = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float rbc2 = 1.f / sqrtf(1.f - powf(beta2, (float)step));
  SdProfScope prof(SD_K_OPTIM, 30.0 * n, (hipStream_t)stream);
  hipLaunchKernelGGL(adamw_f32_shadow_kernel, dim3(nb < 1 ? 1 : nb), dim3(256), 0, (hipStream_t)stream, param,
                     (const bf16*)grad, exp_avg, exp_avg_sq, (bf16*)shadow, (bf16*)shadow_scaled, scale, n4, lr, beta1,
                     beta2, eps, weight_decay, bc1, rbc2, grad_sumsq, max_grad_norm);
  SD_CHECK_LAUNCH();
  return 0;
}
