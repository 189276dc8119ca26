// bf16 MFMA GEMM for gfx950:  C[M,N] = op(A) . op(B) (+ R), fp32 accumulate, bf16 in/out.
//
// Replaces the cuBLAS/ATen GEMMs behind every nn.Linear of the HF Qwen3 decoder the reference
// calls (train.py:54, train.py:63-69; HF modeling_qwen3.py:81-83, 252-254, 279, 441) and their
// autograd backward.  Three operand-layout combinations cover forward and backward without any
// transposed copy in HBM:
//   NT  (ta=0,tb=0)  Y[M,N]  = X[M,K]   . W[N,K]^T      forward linear (torch weight layout [out,in])
//   NN  (ta=0,tb=1)  dX[M,K] = dY[M,N]  . W[N,K]        (B stored [k][n])
//   TN  (ta=1,tb=1)  dW[N,K] = dY[M,N]^T . X[M,K]       (A stored [k][m], B stored [k][n])
//
// Two kernels.  gemm_bf16_kernel: tile BM x 128 x 64 (BM = 256 / 128 / 64, picked per shape so that the grid covers
// the 256 CUs), 4 or 8 waves as (BM/64 or 2) x 2, each wave a (BM / rows-of-waves) x 64 block of
// v_mfma_f32_16x16x32_bf16; NST-deep LDS ring, the wait for tile t is a COUNTED s_waitcnt vmcnt that leaves the
// younger tiles in flight across a raw s_barrier.  gemm_stag_kernel: 256 x 128 x 64, 8 waves, the two 4-wave halves
// run half a K-step apart (one loads while the other computes) -- the dominant kernel of the step.
// Common to both: operand tiles go HBM -> LDS by 16-byte LDS-DMA through buffer descriptors
// (buffer_load_dwordx4 ... lds: loop-invariant per-lane offsets, scalar K advance, hardware range check = zero
// fill; a pointer-based path with a zero page remains for K % 64 != 0 with two K-contiguous operands); the LDS image
// is lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE address and undone on the fragment
// read.  K-contiguous operands are read with ds_read_b128, operands whose contraction index is the slow one with
// ds_read_b64_tr_b16 (hardware transpose, issued as asm so hipcc does not drain the DMA in front of it).  The
// accumulator is produced transposed (mfma(Bfrag, Afrag)) so a lane owns 4 consecutive n; it is staged through LDS
// as fp32 and written out in full 16-byte row pieces with the epilogue fused (+ residual / accumulate).  Split-K
// (contraction over the vocabulary): every K slice writes an fp32 slab; a reduce kernel -- or the consumer itself
// (sd_rmsnorm_bwd_slabs) -- sums the slabs in a fixed order.
#include <stdlib.h>
#include "sd_common.cuh"
#include "../../include/sd_hip.h"
#include "sd_prof.h"
#include "sd_debug.h"

extern "C" __device__ __attribute__((aligned(256))) unsigned char sd_zero_page[1024] = {0};

namespace {

constexpr int BN = 128, BK = 64;

// XOR applied to the 32-byte chunk index of a transposed-operand tile row k ([64 k][ROWS] image)
template <int ROWS> SD_DEV int swz_t(int k) {
  if constexpr (ROWS >= 128) return (k & 3) | (((k >> 3) & 1) << 2);  // >= 8 chunks / row: XOR the low 3 chunk bits
  else return ((k >> 1) & 1) | (((k >> 3) & 1) << 1);                 // 4 chunks / 128-byte row, 2 rows per bank row
}

// Stage one ROWS x 64 operand tile into LDS.  TX=false: operand stored [rows][K] (K contiguous).
// TX=true: operand stored [K][rows] (rows contiguous).
template <bool TX, int ROWS, int NW>
SD_DEV void stage_tile(const bf16* __restrict__ g, long ld, int row0, int k0, int row_lim, int k_lim, char* lds_tile,
                       int w, int lane) {
  constexpr int NI = ROWS / (8 * NW);  // 1 KiB wave-issues per wave
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int p = (w * NI + i) * 64 + lane;
    const bf16* src;
    if constexpr (!TX) {
      const int r = p >> 3, s = p & 7, c = s ^ (r & 7);
      const int gr = row0 + r, gk = k0 + c * 8;
      src = g + (long)gr * ld + gk;
      if (gr >= row_lim || gk >= k_lim) src = (const bf16*)(sd_zero_page + lane * 16);
    } else {
      constexpr int UPR = ROWS / 8;  // 16-byte units per k-row
      const int k = p / UPR, u = p % UPR;
      const int ch = (u >> 1) ^ swz_t<ROWS>(k);
      const int gc = row0 + ch * 16 + (u & 1) * 8, gk = k0 + k;
      src = g + (long)gk * ld + gc;
      if (gk >= k_lim || gc >= row_lim) src = (const bf16*)(sd_zero_page + lane * 16);
    }
    glds16(src, lds_tile + (w * NI + i) * 1024);
  }
}

// Fast staging: buffer_load_dwordx4 ... lds through a buffer descriptor.  The per-lane byte offset of
// every DMA piece is loop-invariant (computed once, kept in VGPRs), the K advance is one scalar add,
// and the hardware range check returns zeros past the end of the operand -- no per-step address
// arithmetic, compares or exec-mask juggling (the checked path costs ~90 SALU + ~30 VALU per K-step).
template <bool TX, int ROWS, int NW>
struct FastStage {
  static constexpr int NI = ROWS / (8 * NW);
  int voff[NI];
  int first_piece;
#if defined(__HIP_DEVICE_COMPILE__)  // the descriptor type only exists in the device pass; the host pass only needs the stub
  __amdgpu_buffer_rsrc_t rsrc;
#endif
  long kstep;  // bytes per k element step of BK
  // hi_delta: rows 64.. of the tile come from row0 + r + hi_delta (EPI 3: the up-projection rows)
  // piece0 >= 0: this wave stages the NI consecutive 1 KiB pieces piece0 .. piece0+NI-1 of the tile (default: w*NI ..)
  SD_DEV void init(const bf16* g, long ld, int row0, unsigned num_bytes, int w, int lane, int hi_delta = 0,
                   int piece0 = -1) {
#if defined(__HIP_DEVICE_COMPILE__)
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, (int)num_bytes, 0x00020000);
#endif
    if (piece0 < 0) piece0 = w * NI;
    first_piece = piece0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int p = (piece0 + i) * 64 + lane;
      if constexpr (!TX) {
        const int r = p >> 3, s = p & 7, c = s ^ (r & 7);
        voff[i] = (int)((((long)(row0 + r + (r >= 64 ? hi_delta : 0))) * ld + c * 8) * 2);
      } else {
        constexpr int UPR = ROWS / 8;
        const int k = p / UPR, u = p % UPR;
        const int ch = (u >> 1) ^ swz_t<ROWS>(k);
        voff[i] = (int)((((long)k) * ld + row0 + ch * 16 + (u & 1) * 8) * 2);
      }
    }
    kstep = TX ? ld * 2 : 2;
  }
  SD_DEV void issue(int k0, char* lds_tile, int w) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const int soff = (int)(k0 * kstep);
#pragma unroll
    for (int i = 0; i < NI; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (SD_LDS void*)(lds_tile + (first_piece + i) * 1024), 16, voff[i], soff, 0, 0);
#endif
  }
};

// Fragment of 16 rows x 32 k for v_mfma_f32_16x16x32_bf16: lane l holds row (l&15), k = 8(l>>4)+j.
// K-contiguous operand image: one ds_read_b128 (transposed operands: load_frags_tr below).
template <bool TX, int ROWS>
SD_DEV bf16x8 load_frag(const char* lds_tile, int row16_base, int kk, int lane) {
  static_assert(!TX, "transposed operands are read by load_frags_tr");
  const int r = row16_base + (lane & 15);
  const int c = kk * 4 + (lane >> 4);
  return *(const bf16x8*)(lds_tile + ((r * 8 + (c ^ (r & 7))) << 4));
}

// Transposed-operand fragments for one kk step, NF fragments (16 rows each from row_base): issues the 2*NF
// ds_read_b64_tr_b16 as asm (see lds_tr16_pair_asm) into raw[], waits once, then packs.
template <int ROWS, int NF>
SD_DEV void load_frags_tr(const char* lds_tile, int row_base, int kk, int lane, bf16x8 (&out)[NF]) {
  static_assert(NF == 2 || NF == 4, "NF");
  sd_u64 raw[2 * NF];
  const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
  const int k0 = kk * 32 + 8 * g + q, k1 = k0 + 4;
  const unsigned base = lds_addr(lds_tile);
  const unsigned r0 = base + k0 * (ROWS * 2) + 8 * pp, r1 = base + k1 * (ROWS * 2) + 8 * pp;
  const int s0 = swz_t<ROWS>(k0), s1 = swz_t<ROWS>(k1);
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int ch = (row_base >> 4) + f;
    lds_tr16_pair_asm(raw[2 * f], raw[2 * f + 1], r0 + ((ch ^ s0) << 5), r1 + ((ch ^ s1) << 5));
  }
  if constexpr (NF == 2) lds_tr_wait4(raw); else lds_tr_wait8(raw);
#pragma unroll
  for (int f = 0; f < NF; ++f) out[f] = cat8_u64(raw[2 * f], raw[2 * f + 1]);
}

// ---- 32-deep stage images (rows of 64 B = 4 chunks): gemm_p256_kernel
constexpr int P2_BK = 32;
// chunk XOR of image row r: ds_read_b128 serves lanes in the groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (rows
// {0-3,12-15} at k-chunk c with rows {4-11} at c^1): with g(r>>2) = 0,3,2,1 the 16 lanes of every group fall on 16
// different 16-byte slots of the 256-byte bank row (the plain (r>>2)&3 gives 2-way conflicts: 0.9 instead of 0.5 us
// per K-step).
SD_DEV int swz32(int r) { return (0 - (r >> 2)) & 3; }
SD_DEV bf16x8 load_frag32(const char* tile, int row16_base, int lane) {  // 16 rows x 32 k: lane l = row l&15, k 8(l>>4)..
  const int r = row16_base + (lane & 15), c = lane >> 4;
  return *(const bf16x8*)(tile + r * 64 + ((c ^ swz32(r)) << 4));
}

// Extra operands of the fused epilogues.
//   EPI 3 (gate|up GEMM + SwiGLU, HF:81-83): the B tile of column-tile tn is gate rows [64tn,64tn+64) followed by up
//          rows [I+64tn, ...), so a 128-column C tile holds gate and up of the SAME 64 outputs; act = silu(gate)*up
//          is written to out2 [M,I], gate|up (needed by the backward) to C [M,2I] when C != nullptr.
//   EPI 4 (q|k|v GEMM + per-head RMSNorm + rotate-half RoPE, HF:252-257,121-170): one 128-column tile is exactly one
//          head; raw q|k|v go to C, normalised + rotated q|k heads to out2 [M,(Hq+Hkv)*128].
//   EPI 5 (down-projection dX GEMM + SwiGLU backward, HF:81-83 run backwards): the tile is d(act); g0 = gate|up of the
//          forward [M,2I] (row stride ld2), out2 = d(gate|up) [M,2I]; d(act) itself is not stored.
//   Folded RMSNorm (round 4; the frozen teacher, HF:59-64 + the projection behind it): the norm's gain is folded into the
//   weight rows once at load (W' = W.diag(g)), so y = rstd (.) (x W'^T) and the norm needs no pass of its own:
//     ssq_out (EPI 1): the residual epilogue also leaves, per row and 128-column tile, the sum of squares of the bf16
//             values it stores: ssq_out[tile * M + row] (fp32, TILE-major [ssq_n][M], ssq_n = N / 128 <= 16: a 64-byte
//             line holds 16 rows of ONE tile, i.e. is written by one workgroup -- row-major, 16 workgroups on different
//             XCDs each wrote 4 bytes of every line and the launch took +10 % at 32 768 rows);
//     ssq_in  (EPI 3 / 4): the consumer sums a row's ssq_n partials in a fixed order, rstd = rsqrt(sum * inv_h + eps_rs),
//             and scales the accumulator row by it before anything else happens to it.  The partials are FETCHED before
//             the K loop and only USED in the epilogue: a use in the prologue made every workgroup wait for them before
//             its first DMA piece went out (+11 us per tile with the memory system busy, config 5).
struct EpiArgs {
  bf16* out2;
  long ld2;
  const bf16 *g0, *g1, *cos_t, *sin_t;
  int T, Hq, Hkv, I;
  float eps;
  float* ssq_out;
  const float* ssq_in;
  int ssq_n;
  float inv_h, eps_rs;
};

// fp32 C tile in LDS ([BM][128], 16-byte chunks XOR-swizzled by row) -> global, 8 columns per thread.
// Global operands of the epilogue (residual rows for EPI 1; cos / sin rows for EPI 4) are loaded into registers
// BEFORE the K loop (epi_preload): with one or two workgroups per CU nothing else would hide their latency
// between the last MFMA and the stores (measured: 7-9 us per launch when loaded inside the epilogue).
template <int EPI, int BM, int NTHR>
struct EpiPre {
  static constexpr int IT = BM * 16 / NTHR;
  bf16x8 a[(EPI == 1 || EPI == 4 || EPI == 5 || EPI == 6) ? IT : 1];
  bf16x8 b[(EPI == 4 || EPI == 5) ? IT : 1];
  bf16x8 gain;
  float rs[(EPI == 3 || EPI == 4) ? IT : 1];  // folded RMSNorm: this thread's partial sum of squares of its row (raw)
};

template <int EPI, int BM, int NTHR>
SD_DEV void epi_preload(EpiPre<EPI, BM, NTHR>& pre, const bf16* R, const EpiArgs& ea, int M, int N, long ldr, int m0,
                        int n0, int tn) {
  if constexpr (EPI == 1 || EPI == 4 || EPI == 5 || EPI == 6) {
#pragma unroll
    for (int it = 0; it < BM * 16 / NTHR; ++it) {
      const int q = it * NTHR + threadIdx.x;
      const int m = q >> 4, oc = q & 15;
      const int gm = m0 + m, gn = n0 + oc * 8;
      if constexpr (EPI == 1) {
        const bool ok = gm < M && gn < N;
        bf16x8 z = {};
        pre.a[it] = ok ? *(const bf16x8*)(R + (long)gm * ldr + gn) : z;
      } else if constexpr (EPI == 6) {  // the attention output O at this (row, 8 columns of head tn)
        const bool ok = gm < M && gn < N;
        bf16x8 z = {};
        pre.a[it] = ok ? *(const bf16x8*)(ea.g0 + (long)gm * ea.ld2 + gn) : z;
      } else if constexpr (EPI == 5) {  // gate and up of the forward at this (row, 8 columns)
        const bool ok = gm < M && gn < N;
        bf16x8 z = {};
        pre.a[it] = ok ? *(const bf16x8*)(ea.g0 + (long)gm * ea.ld2 + gn) : z;
        pre.b[it] = ok ? *(const bf16x8*)(ea.g0 + (long)gm * ea.ld2 + ea.I + gn) : z;
      } else {
        const int gmc = gm < M ? gm : M - 1;
        const int t = gmc % ea.T;
        pre.a[it] = *(const bf16x8*)(ea.cos_t + (long)t * 128 + oc * 8);
        pre.b[it] = *(const bf16x8*)(ea.sin_t + (long)t * 128 + oc * 8);
      }
    }
    if constexpr (EPI == 4) pre.gain = *(const bf16x8*)((tn < ea.Hq ? ea.g0 : ea.g1) + (threadIdx.x & 15) * 8);
  }
  if constexpr (EPI == 3 || EPI == 4) {
    // the 16 threads of a row (one DPP row) fetch one partial each; they are summed in the epilogue (row_scale)
#pragma unroll
    for (int it = 0; it < BM * 16 / NTHR; ++it) {
      float part = 0.f;
      if (ea.ssq_in) {  // kernel-uniform
        const int q = it * NTHR + threadIdx.x;
        const int gm = m0 + (q >> 4), oc = q & 15;
        if (oc < ea.ssq_n) part = ea.ssq_in[(long)oc * M + (gm < M ? gm : M - 1)];
      }
      pre.rs[it] = part;
    }
  }
}

// rstd of this thread's row from the 16 partials its DPP row holds (fixed butterfly order); every lane must be active
SD_DEV float row_scale(float part, const EpiArgs& ea) { return rsqrtf(row16_sum(part) * ea.inv_h + ea.eps_rs); }

template <int EPI, int BM, int NTHR>
SD_DEV void write_out(const float* cs, bf16* C, const EpiPre<EPI, BM, NTHR>& pre, float* slabs, const EpiArgs& ea, int M,
                      int N, long ldc, int m0, int n0, int tn, int slice) {
#pragma unroll
  for (int it = 0; it < BM * 16 / NTHR; ++it) {
    const int q = it * NTHR + threadIdx.x;
    const int m = q >> 4, oc = q & 15;
    const int gm = m0 + m, gn = n0 + oc * 8;
    const bool ok = gm < M && gn < N;
    const f32x4 lo = *(const f32x4*)(cs + m * 128 + (((2 * oc) ^ (m & 15)) << 2));
    const f32x4 hi = *(const f32x4*)(cs + m * 128 + (((2 * oc + 1) ^ (m & 15)) << 2));
    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    if constexpr (EPI == 2) {
      if (ok) {
        float* dst = slabs + ((long)slice * M + gm) * N + gn;
        *(f32x4*)dst = lo;
        *(f32x4*)(dst + 4) = hi;
      }
    } else if constexpr (EPI == 3) {
      // columns 0..63 = gate, 64..127 = up of outputs 64*tn + (0..63); threads oc < 8 own one 8-wide output chunk
      const int oc2 = (oc & 7) + 8;
      const f32x4 ulo = *(const f32x4*)(cs + m * 128 + (((2 * oc2) ^ (m & 15)) << 2));
      const f32x4 uhi = *(const f32x4*)(cs + m * 128 + (((2 * oc2 + 1) ^ (m & 15)) << 2));
      float u[8] = {ulo[0], ulo[1], ulo[2], ulo[3], uhi[0], uhi[1], uhi[2], uhi[3]};
      if (ea.ssq_in) {
        const float rsc = row_scale(pre.rs[it], ea);
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] *= rsc; u[e] *= rsc; }
      }
      const int col = tn * 64 + oc * 8;
      if (oc < 8 && gm < M && col < ea.I) {
        bf16x8 gb, ub, ab;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          gb[e] = (bf16)v[e];
          ub[e] = (bf16)u[e];
          const float gf = (float)gb[e], uf = (float)ub[e];
          ab[e] = (bf16)(gf / (1.f + __expf(-gf)) * uf);
        }
        if (C) {
          *(bf16x8*)(C + (long)gm * ldc + col) = gb;
          *(bf16x8*)(C + (long)gm * ldc + ea.I + col) = ub;
        }
        *(bf16x8*)(ea.out2 + (long)gm * ea.ld2 + col) = ab;
      }
    } else if constexpr (EPI == 5) {
      // SwiGLU backward on d(act) = this tile (rounded to bf16 first, as the unfused pair stores it): d(gate), d(up)
      if (ok) {
        bf16x8 dg, du;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = (float)(bf16)v[e], g = (float)pre.a[it][e], u = (float)pre.b[it][e];
          const float sg = 1.f / (1.f + __expf(-g));
          du[e] = (bf16)(d * g * sg);
          dg[e] = (bf16)(d * u * sg * (1.f + g * (1.f - sg)));
        }
        *(bf16x8*)(ea.out2 + (long)gm * ea.ld2 + gn) = dg;
        *(bf16x8*)(ea.out2 + (long)gm * ea.ld2 + ea.I + gn) = du;
      }
    } else if constexpr (EPI == 6) {
      // o-projection dX: the tile is d(attention output) of head tn; beside storing it, delta = rowsum(dO * O) of that
      // head falls out (the 16 threads of a row hold its 128 columns) -- same arithmetic and order as attn_delta_kernel
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
      if (ok) *(bf16x8*)(C + (long)gm * ldc + gn) = o;
      float sdel = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) sdel += (float)o[e] * (float)pre.a[it][e];
      sdel = row16_sum(sdel);
      if (oc == 0 && gm < M) {
        const int bb = gm / ea.T, t = gm - bb * ea.T;
        ((float*)ea.out2)[((long)bb * ea.Hq + tn) * ea.T + t] = sdel;
      }
    } else if constexpr (EPI == 4) {
      bf16x8 raw;
      if (ea.ssq_in) {
        const float rsc = row_scale(pre.rs[it], ea);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= rsc;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) raw[e] = (bf16)v[e];
      if (ok) *(bf16x8*)(C + (long)gm * ldc + gn) = raw;
      if (tn < ea.Hq + ea.Hkv) {  // block-uniform: q or k head -> RMSNorm over the 128 columns, then RoPE
        float f[8], ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { f[e] = (float)raw[e]; ss += f[e] * f[e]; }
        ss = row16_sum(ss);
        const float rs = rsqrtf(ss * (1.f / 128.f) + ea.eps);
        const bf16x8 gv = pre.gain, cv = pre.a[it], sv = pre.b[it];
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float nrm = (float)(bf16)((float)gv[e] * (float)(bf16)(f[e] * rs));
          const float pr = row16_xor8(nrm);
          const float rot = (oc < 8) ? -pr : pr;
          o[e] = (bf16)(nrm * (float)cv[e] + rot * (float)sv[e]);
        }
        if (ok) *(bf16x8*)(ea.out2 + (long)gm * ea.ld2 + gn) = o;
      }
    } else {
      if constexpr (EPI == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)pre.a[it][e];
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
      if (ok) *(bf16x8*)(C + (long)gm * ldc + gn) = o;
      if constexpr (EPI == 1) {
        if (ea.ssq_out) {  // kernel-uniform; every lane of the row takes part in the DPP sum
          float ss = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float f = ok ? (float)o[e] : 0.f; ss += f * f; }
          ss = row16_sum(ss);
          if (oc == 0 && gm < M) ea.ssq_out[(long)tn * M + gm] = ss;
        }
      }
    }
  }
}

// EPI: 0 = bf16 out, 1 = bf16 out + residual, 2 = fp32 slab out (split-K)
// Tile order inside the launch.  Workgroups that are resident together on one XCD (32 CUs behind one 4 MiB L2) should
// share operand panels: with the plain column-major order an XCD's 32 consecutive tiles are 32 row tiles of ONE column
// (for BM = 64: 272 KiB of unique operand bytes per K-step, L2 hit 65 %); walking `group_m` row tiles down, then
// across all columns, makes them a group_m x (32/group_m) block (8 x 4: 128 KiB, hit 83 %).  The kernels are bound
// by the L2 -> LDS rate (45-48 GB/s per CU at every tile shape), so the misses that go to the Infinity Cache count.
SD_DEV void tile_coords(int tile, int tiles_m, int tiles_n, int group_m, int& tm, int& tn) {
  const int per_group = group_m * tiles_n;
  const int g = tile / per_group;
  const int r = tile - g * per_group;
  const int g0 = g * group_m;
  const int gh = min(group_m, tiles_m - g0);
  tn = r / gh;
  tm = g0 + (r - tn * gh);
}

// Which (tile, K slice) a workgroup of a (tiles, slices) grid takes.  The hardware deals workgroups to the 8 XCDs
// round-robin by LINEAR id (x fastest), so remapping blockIdx.x alone -- as round 2 did -- leaves every XCD with a few
// tiles of ALL the slices: for the gate|up dX (64 tiles x 4 slices) each 4 MiB L2 then streams 15.7 MB of operand
// panels (272 MB per launch at the fabric against 71 MB algorithmic, profiles/r02_pmc_traffic.json).  Remapping the
// linear id over tiles x slices gives an XCD's 32 resident workgroups ONE slice of a near-square block of tiles
// (6.3 MB per XCD).  One slice: identical to the old order.  Speed only.
SD_DEV void tile_and_slice(int ntiles, int& tile, int& slice) {
  const int g = xcd_remap((int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y, ntiles * (int)gridDim.y);
  slice = g / ntiles;
  tile = g - slice * ntiles;
}

// NST-deep LDS ring: tile t+NST-1 is issued while tile t is computed; the wait for tile t is a COUNTED
// s_waitcnt vmcnt that leaves the NST-2 younger tiles in flight across the (raw) barrier.  Tiles past
// the end of K are still issued (their lanes read the zero page), which keeps the count uniform.
template <int BM, int NST, bool TA, bool TB, int EPI, bool FAST>
__global__ __launch_bounds__(BM == 256 ? 512 : 256, (BM == 256 ? 2 : (NST * (BM + 128) * 128 <= 80 * 1024 ? 2 : 1))) void gemm_bf16_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* C,
                                                        const bf16* R, float* __restrict__ slabs, int M, int N, int K,
                                                        long lda, long ldb, long ldc, long ldr, int tiles_m, int tiles_n,
                                                        int k_tiles_per_split, int group_m, EpiArgs ea) {
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int NW = (BM == 256) ? 8 : 4;          // waves: (NW/2) along M x 2 along N
  constexpr int NTHR = NW * 64;
  constexpr int WROWS = BM / (NW / 2);             // rows of C per wave (64, 64, 32)
  constexpr int MT = WROWS / 16;                   // 16-row MFMA tiles per wave along M
  constexpr int EPI_BYTES = BM * BN * 4;
  constexpr int SMEM = (NST * STAGE > EPI_BYTES) ? NST * STAGE : EPI_BYTES;
  constexpr int LOADS = (BM + BN) / (8 * NW);      // LDS-DMA instructions per wave per tile
  __shared__ __attribute__((aligned(16))) char smem[SMEM];
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int wm = w >> 1, wn = w & 1;
  int tile, slice;
  tile_and_slice(tiles_m * tiles_n, tile, slice);
  int tm, tn;
  tile_coords(tile, tiles_m, tiles_n, group_m, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  EpiPre<EPI, BM, NTHR> pre;
  epi_preload<EPI, BM, NTHR>(pre, R, ea, M, N, ldr, m0, n0, tn);

  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int kt_all = (K + BK - 1) / BK;
  const int kt0 = slice * k_tiles_per_split;
  const int kt1 = min(kt_all, kt0 + k_tiles_per_split);
  const int nk = kt1 - kt0;
  const int k_end = min(K, kt1 * BK);
  FastStage<TA, BM, NW> fa;
  FastStage<TB, BN, NW> fb;
  if constexpr (FAST) {
    fa.init(A, lda, m0, (unsigned)((TA ? ((long)(K - 1) * lda + M) : ((long)(M - 1) * lda + K)) * 2), w, lane);
    fb.init(B, ldb, EPI == 3 ? tn * 64 : n0, (unsigned)((TB ? ((long)(K - 1) * ldb + N) : ((long)(N - 1) * ldb + K)) * 2),
            w, lane, EPI == 3 ? ea.I - 64 : 0);
  }
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) {
    if constexpr (FAST) {
      fa.issue((kt0 + s) * BK, smem + s * STAGE, w);
      fb.issue((kt0 + s) * BK, smem + s * STAGE + A_BYTES, w);
    } else {
      stage_tile<TA, BM, NW>(A, lda, m0, (kt0 + s) * BK, M, k_end, smem + s * STAGE, w, lane);
      stage_tile<TB, BN, NW>(B, ldb, n0, (kt0 + s) * BK, N, k_end, smem + s * STAGE + A_BYTES, w, lane);
    }
  }
  int cur_i = 0, nxt_i = NST - 1;
  for (int t = 0; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * LOADS) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
      char* nxt = smem + nxt_i * STAGE;
      if constexpr (FAST) {
        fa.issue((kt0 + t + NST - 1) * BK, nxt, w);
        fb.issue((kt0 + t + NST - 1) * BK, nxt + A_BYTES, w);
      } else {
        stage_tile<TA, BM, NW>(A, lda, m0, (kt0 + t + NST - 1) * BK, M, k_end, nxt, w, lane);
        stage_tile<TB, BN, NW>(B, ldb, n0, (kt0 + t + NST - 1) * BK, N, k_end, nxt + A_BYTES, w, lane);
      }
    }
    const char* cur = smem + cur_i * STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[MT], bfr[4];
      if constexpr (TA) {
        load_frags_tr<BM, MT>(cur, wm * WROWS, kk, lane, af);
      } else {
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = load_frag<TA, BM>(cur, wm * WROWS + i * 16, kk, lane);
      }
      if constexpr (TB) {
        load_frags_tr<BN, 4>(cur + A_BYTES, wn * 64, kk, lane, bfr);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = load_frag<TB, BN>(cur + A_BYTES, wn * 64 + j * 16, kk, lane);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(bfr[j], af[i], acc[i][j]);  // D[n][m]: lane owns 4 consecutive n
    }
    cur_i = (cur_i + 1 == NST) ? 0 : cur_i + 1;
    nxt_i = (nxt_i + 1 == NST) ? 0 : nxt_i + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the (zero-page) tail tiles before LDS is reused
  __syncthreads();

  // Epilogue: fp32 tile -> LDS (XOR-swizzled 16-byte chunks), then coalesced rows out.
  float* cs = (float*)smem;  // [BM][128] fp32
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = wm * WROWS + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cidx = wn * 16 + j * 4 + (lane >> 4);
      *(f32x4*)(cs + m * 128 + ((cidx ^ (m & 15)) << 2)) = acc[i][j];
    }
  }
  __syncthreads();
  write_out<EPI, BM, NTHR>(cs, C, pre, slabs, ea, M, N, ldc, m0, n0, tn, slice);
}

// ---------------------------------------------------------------------------------------------------
// Staggered 256x128x64 kernel (8 waves).  The L1/TA path moves 64 B/clk/CU, which for a 128x128 tile is
// as much time as its MFMAs take, and waves of one workgroup that all issue their LDS-DMA right after
// the same barrier and then all compute cannot overlap the two.  Here the two 4-wave halves of the
// workgroup (one wave of each half per SIMD) run half a K-step apart: while one half is in its LOAD
// phase (issue the DMA of tile t+2, read the fragments of tile t from LDS) the other half is in its
// COMPUTE phase (32 MFMAs on fragments already in registers), then they swap at the next barrier.
//   LOAD(t):    issue tile t+2 -> stage (t+2)%3 | ds_read tile t | vmcnt(6): my tile t+1 landed | lgkmcnt(0)
//   COMPUTE(t): 32 x v_mfma_f32_16x16x32_bf16
// Hazards (phase p = 2t for half 0, 2t+1 for half 1; a barrier separates consecutive phases):
//   RAW  tile x is first read in phase 2x; every wave retires its own tile-x DMA with the counted vmcnt at
//        the end of its LOAD(x-1) (phases 2x-2 / 2x-1) and then passes a barrier.
//   WAR  stage (x-1)%3 is re-filled from phase 2x on; its last reads (tile x-1) were issued in phases
//        2x-2 / 2x-1 and completed (lgkmcnt(0)) before those phases' closing barriers.
template <bool TA, bool TB, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_stag_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* C,
                                                           const bf16* R, float* __restrict__ slabs, int M, int N, int K,
                                                           long lda, long ldb, long ldc, long ldr, int tiles_m,
                                                           int tiles_n, int k_tiles_per_split, int group_m, EpiArgs ea) {
  constexpr int BM = 256, NW = 8, NST = 3;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;  // 48 KiB
  constexpr int LOADS = (BM + BN) / (8 * NW);                                             // 6 per wave per tile
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];                         // 144 KiB (epilogue: 128 KiB)
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int wm = w >> 1, wn = w & 1;
  const int half = w >> 2;  // waves 0-3 / 4-7: one of each per SIMD
  int tile, slice;
  tile_and_slice(tiles_m * tiles_n, tile, slice);
  int tm, tn;
  tile_coords(tile, tiles_m, tiles_n, group_m, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  EpiPre<EPI, 256, 512> pre;
  epi_preload<EPI, 256, 512>(pre, R, ea, M, N, ldr, m0, n0, tn);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int kt_all = (K + BK - 1) / BK;
  const int kt0 = slice * k_tiles_per_split;
  const int kt1 = min(kt_all, kt0 + k_tiles_per_split);
  const int nk = kt1 - kt0;
  FastStage<TA, BM, NW> fa;
  FastStage<TB, BN, NW> fb;
  fa.init(A, lda, m0, (unsigned)((TA ? ((long)(K - 1) * lda + M) : ((long)(M - 1) * lda + K)) * 2), w, lane);
  fb.init(B, ldb, EPI == 3 ? tn * 64 : n0, (unsigned)((TB ? ((long)(K - 1) * ldb + N) : ((long)(N - 1) * ldb + K)) * 2), w,
          lane, EPI == 3 ? ea.I - 64 : 0);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    fa.issue((kt0 + s) * BK, smem + s * STAGE, w);
    fb.issue((kt0 + s) * BK, smem + s * STAGE + A_BYTES, w);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (half == 1) __builtin_amdgcn_s_barrier();  // second half runs one phase behind

  int cur_i = 0, nxt_i = 2;
  for (int t = 0; t < nk; ++t) {
    // ---- LOAD(t)
    {
      char* nxt = smem + nxt_i * STAGE;
      fa.issue((kt0 + t + 2) * BK, nxt, w);
      fb.issue((kt0 + t + 2) * BK, nxt + A_BYTES, w);
    }
    const char* cur = smem + cur_i * STAGE;
    bf16x8 af[2][4], bfr[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if constexpr (TA) {
        load_frags_tr<BM, 4>(cur, wm * 64, kk, lane, af[kk]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[kk][i] = load_frag<TA, BM>(cur, wm * 64 + i * 16, kk, lane);
      }
      if constexpr (TB) {
        load_frags_tr<BN, 4>(cur + A_BYTES, wn * 64, kk, lane, bfr[kk]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[kk][j] = load_frag<TB, BN>(cur + A_BYTES, wn * 64 + j * 16, kk, lane);
      }
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- COMPUTE(t)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(bfr[kk][j], af[kk][i], acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    cur_i = (cur_i == 2) ? 0 : cur_i + 1;
    nxt_i = (nxt_i == 2) ? 0 : nxt_i + 1;
  }
  if (half == 0) __builtin_amdgcn_s_barrier();  // re-align the halves
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  float* cs = (float*)smem;  // [256][128] fp32
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = wm * 64 + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cidx = wn * 16 + j * 4 + (lane >> 4);
      *(f32x4*)(cs + m * 128 + ((cidx ^ (m & 15)) << 2)) = acc[i][j];
    }
  }
  __syncthreads();
  write_out<EPI, 256, 512>(cs, C, pre, slabs, ea, M, N, ldc, m0, n0, tn, slice);
}

// ---------------------------------------------------------------------------------------------------
// Persistent form of the staggered 256x128x64 kernel for launches with more tiles than CUs (lm_head forward and dW,
// the teacher's gate|up: 3-20 tiles per CU).  Measured on the one-tile-per-workgroup kernel: a K-step takes 0.79 us
// (1.36 PFLOP/s chip-wide) but every tile costs another ~8 us -- descriptor set-up and the first two stage loads in
// front, LDS-staged write-out, workgroup retire and the launch of the next one behind -- 38 % of a K = 1024 tile.
// Here a workgroup walks its tiles (the same per-XCD order as the other kernels) as ONE K stream: the LDS-DMA runs
// two K-steps ahead ACROSS tile boundaries (the tile origin is folded into the scalar offset of the buffer load),
// and a wave that finishes a tile stores its 64x64 block straight from the accumulators (8-byte pieces, merged in L2)
// -- LDS stays with the ring, nothing waits for the next tile's operands.  12 waves: 8 compute (the two staggered
// halves of gemm_stag_kernel) + 4 producers that issue every LDS-DMA and do all the waiting on memory.
//   EPI 0: C = A.B          EPI 3: SwiGLU (B tile = 64 gate rows | 64 up rows; each wave reads 32 + 32 of them, so gate
//   and up of the same outputs sit in the same lane: act = silu(gate) * up in registers, out2 [M, I]).
// (A 128 x 128 form of this kernel with a 4-stage ring was measured on the N = hidden shapes: no faster than the
// 4-wave gemm_bf16_kernel<128, 3>, so it is not kept.  Round 2 built, verified bit-identical and measured two
// single-barrier forms on the gate|up / lm_head shapes -- 8 waves with whole K-steps of fragments double-buffered in
// registers and self-issued DMA, halves in opposite orders (teacher gate|up 107.6 us, lm_head 983 us), and 8 + 4 waves
// with half-step fragment double buffering over a ring of six 32-deep half-stages (123.8 / 1 115 us) -- against this
// kernel's 97.0 / 892 us: not kept either)
template <int MT, bool TA, bool TB, int EPI>
__global__ __launch_bounds__(768) void gemm_pstag_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* C,
                                                         const bf16* R, int M, int N, int K, long lda, long ldb, long ldc,
                                                         long ldr, int tiles_m, int tiles_n, int group_m, EpiArgs ea) {
  static_assert(MT == 4 && (EPI == 0 || EPI == 3), "256 x 128 tile; plain or SwiGLU epilogue");
  constexpr int BM = 64 * MT, NW = 8, NPROD = 4, NST = 3, DEPTH = NST - 1;
  constexpr int WR = 16 * MT;                                                             // rows of C per compute wave
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;  // 48 / 32 KiB
  // LDS-DMA issue is what paces the K loop: one buffer_load ... lds costs the issuing wave ~140 cycles, serially, so
  // the 48 pieces of a stage took the 4 producers 12 x 140 = 1 650 cycles -- exactly the measured 0.81 us per K-step
  // (and the 8 self-issuing waves of gemm_stag_kernel 6 x 140 + reads + 512 of MFMA, the same 0.81 us).  Spread over
  // more waves the pieces overlap until the CU's address path (16 cycles a piece) or the MFMAs (2 x 512) bound the
  // step: the producers keep B and the first half of A (8 pieces each, ~1 100 cycles), every compute wave adds 2
  // pieces of the second half of A to its LOAD phase (2 x 140 + 16 reads, still under the partner's 512 of MFMA).
  constexpr int A_PIECES = BM / 8, B_PIECES = BN / 8;           // 1 KiB pieces per stage: 32 + 16
  constexpr int CW = (A_PIECES / 2) / NW;                       // A pieces per compute wave and stage: 2
  constexpr int PA = (A_PIECES / 2) / NPROD, PB = B_PIECES / NPROD;  // per producer wave and stage: 4 + 4
  constexpr int LOADS = PA + PB;
  constexpr int PATCH = 2048;                                                             // per compute wave
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE + NW * PATCH];            // ring + store patches <= 160 KiB
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int ntiles = tiles_m * tiles_n;
  const int nk = (K + BK - 1) / BK;
  // my tiles: xcd_remap(blockIdx.x + k * gridDim.x) (gridDim.x is a multiple of 8, or ntiles itself)
  const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_tiles * nk;  // K-steps of this workgroup
#ifdef SD_STAMPS
  // DIAGNOSTIC BUILD ONLY (make stamps): s_memtime at the phase boundaries of K-steps 4..19 of workgroup 0, kept in the
  // wave's idle store patch (LDS) and dumped to ea.cos_t right after the window: [wave 0..11][step 0..15][point 0..5]
  unsigned long long* const stamp_out = (unsigned long long*)ea.cos_t;
  const bool stamping = stamp_out && blockIdx.x == 0;
#define SD_STAMP(G, PT)                                                                                   \
  do {                                                                                                    \
    if (stamping && (G) >= 4 && (G) < 20 && lane == 0)                                                    \
      ((unsigned long long*)(smem + NST * STAGE + (w & 7) * PATCH + (w >= NW ? 1024 : 0)))[((G) - 4) * 6 + (PT)] = \
          __builtin_amdgcn_s_memtime();                                                                   \
  } while (0)
#define SD_STAMP_DUMP(G)                                                                                  \
  do {                                                                                                    \
    if (stamping && (G) == 20 && lane < 48) {                                                             \
      const unsigned long long* src = (const unsigned long long*)(smem + NST * STAGE + (w & 7) * PATCH + (w >= NW ? 1024 : 0)); \
      stamp_out[w * 96 + lane] = src[lane];                                                               \
      stamp_out[w * 96 + 48 + lane] = src[48 + lane];                                                     \
    }                                                                                                     \
  } while (0)
#else
#define SD_STAMP(G, PT) do { } while (0)
#define SD_STAMP_DUMP(G) do { } while (0)
#endif
  auto origin = [&](int idx, int& tm, int& tn) {
    const int t = xcd_remap((int)blockIdx.x + idx * (int)gridDim.x, ntiles);
    tile_coords(t, tiles_m, tiles_n, group_m, tm, tn);
  };

  // ------------------------------------------------------------------ producer waves 8..11: the operand stream
  // They alone issue LDS-DMA and wait on vmcnt, so the 8 compute waves never wait for memory: gfx950 counts loads
  // and stores in ONE in-order counter, and a compute wave that had just stored its finished tile would sit at its
  // next counted wait until those stores were acknowledged.  Barrier count per wave: 1 + 2 per K-step + 1, the same
  // in all three roles.  Hazards (phase 2g = LOAD(g) of half 0, 2g+1 = LOAD(g) of half 1; a barrier between phases):
  //   RAW  K-step g is first read in phase 2g; the producers retired it (counted vmcnt) by the end of phase 2g-2.
  //   WAR  K-step g+DEPTH goes to the stage of K-step g-1, issued in phase 2g; the last reads of g-1 were issued in
  //        phases 2g-2 / 2g-1 and completed (lgkmcnt(0)) before those phases' closing barriers.
  if (w >= NW) {
    const int pw = w - NW;
    FastStage<TA, BM, A_PIECES / PA> fa;  // NI = PA pieces: pieces pw*PA .. of the first half of the A tile
    FastStage<TB, BN, NPROD> fb;
    static_assert(FastStage<TA, BM, A_PIECES / PA>::NI == PA && FastStage<TB, BN, NPROD>::NI == PB, "piece split");
    fa.init(A, lda, 0, (unsigned)((TA ? ((long)(K - 1) * lda + M) : ((long)(M - 1) * lda + K)) * 2), pw, lane, 0, pw * PA);
    fb.init(B, ldb, 0, (unsigned)((TB ? ((long)(K - 1) * ldb + N) : ((long)(N - 1) * ldb + K)) * 2), pw, lane,
            EPI == 3 ? ea.I - 64 : 0);
    int pf_tile = 0, pf_k = 0;
    unsigned pf_a = 0, pf_b = 0;  // byte offsets of the prefetch tile's origin in A and B
    (void)pf_a; (void)pf_b;
    auto pf_set = [&](int idx) {
      int tm = 0, tn = 0;
      if (idx < my_tiles) origin(idx, tm, tn);
      const long m0 = (long)tm * BM, nb0 = (EPI == 3) ? (long)tn * 64 : (long)tn * BN;
      pf_a = (unsigned)((TA ? m0 : m0 * lda) * 2);
      pf_b = (unsigned)((TB ? nb0 : nb0 * ldb) * 2);
    };
    auto pf_issue = [&](char* stage) {
#if defined(__HIP_DEVICE_COMPILE__)
      const int sa = (int)(pf_a + (unsigned)(pf_k * BK) * (unsigned)fa.kstep);
      const int sb = (int)(pf_b + (unsigned)(pf_k * BK) * (unsigned)fb.kstep);
#pragma unroll
      for (int i = 0; i < PA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(fa.rsrc, (SD_LDS void*)(stage + (pw * PA + i) * 1024), 16, fa.voff[i], sa, 0,
                                                 0);
#pragma unroll
      for (int i = 0; i < FastStage<TB, BN, NPROD>::NI; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            fb.rsrc, (SD_LDS void*)(stage + A_BYTES + (pw * FastStage<TB, BN, NPROD>::NI + i) * 1024), 16, fb.voff[i], sb,
            0, 0);
#endif
      if (++pf_k == nk) { pf_k = 0; pf_set(++pf_tile); }
    };
    pf_set(0);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) pf_issue(smem + d * STAGE);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int nxt = DEPTH;
    // Folded RMSNorm (EPI 3 with ea.ssq_in): the 4 x 64 producer lanes are the 256 rows of the tile being COMPUTED.  In the
    // tile's first K-step each lane fetches its row's partial sums of squares (16 loads of 4 B, each 256 contiguous bytes
    // per wave) AHEAD of that step's DMA pieces, so the step's ordinary counted wait covers them (they are
    // older than the pieces it leaves in flight); then it writes rstd into the LDS table of the tile's parity (the store
    // patches, idle in this epilogue).  The compute waves read it at the end of the tile, nk K-steps of barriers later.
    [[maybe_unused]] int ct = 0, ck = 0;  // tile / K-step of the step being computed
    for (int g = 0; g < total; ++g) {
      // phase 2g: K-step g+DEPTH of the stream (steps past the end re-read the first origin and are never used);
      // then everything up to K-step g+1 has landed
      SD_STAMP_DUMP(g);
      SD_STAMP(g, 0);
      [[maybe_unused]] float sq[16];
      bool fold_now = false;
      if constexpr (EPI == 3) {
        fold_now = ea.ssq_in != nullptr && ck == 0;  // workgroup-uniform
        if (fold_now) {
          int tm, tn;
          origin(ct, tm, tn);
          int gm = tm * BM + pw * 64 + lane;
          gm = gm < M ? gm : M - 1;
          // tile-major partials: one load per tile, 64 consecutive rows per wave-instruction (256 B); always 16 loads
          // (tiles past ssq_n re-read the last one and count as 0) so that the step's counted wait sees a fixed number
#pragma unroll
          for (int t = 0; t < 16; ++t) sq[t] = ea.ssq_in[(long)(t < ea.ssq_n ? t : ea.ssq_n - 1) * M + gm];
        }
      }
      pf_issue(smem + nxt * STAGE);
      nxt = (nxt == NST - 1) ? 0 : nxt + 1;
      SD_STAMP(g, 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * LOADS) : "memory");
      if constexpr (EPI == 3) {
        if (fold_now) {
          // the same balanced tree as row16_sum over 16 lanes (row_scale in the one-tile kernels): bit-identical rstd
          float qd[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            float p4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) p4[e] = (4 * c + e < ea.ssq_n) ? sq[4 * c + e] : 0.f;
            qd[c] = (p4[0] + p4[1]) + (p4[2] + p4[3]);
          }
          const float tot = (qd[0] + qd[1]) + (qd[2] + qd[3]);
          ((float*)(smem + NST * STAGE + (ct & 1) * 1024))[pw * 64 + lane] = rsqrtf(tot * ea.inv_h + ea.eps_rs);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (ea.ssq_in && ++ck == nk) { ck = 0; ++ct; }
      }
      SD_STAMP(g, 2);
      __builtin_amdgcn_s_barrier();
      SD_STAMP(g, 3);
      __builtin_amdgcn_s_barrier();  // phase 2g+1: nothing to do
      SD_STAMP(g, 4);
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the tail prefetches before the workgroup retires
    return;
  }

  // ------------------------------------------------------------------ compute waves 0..7
  const int wm = w >> 1, wn = w & 1;
  const int half = w >> 2;
  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // this wave's share of the operand stream: CW pieces of the second half of every A stage, issued in LOAD(g) for
  // K-step g+DEPTH and retired by a counted vmcnt before the barrier that closes the phase (the same RAW / WAR
  // argument as the producers'; a finished tile's stores sit in the same in-order counter, so the first wait after a
  // tile boundary also waits for them)
  FastStage<TA, BM, A_PIECES / CW> fc;
  static_assert(FastStage<TA, BM, A_PIECES / CW>::NI == CW, "piece split");
  fc.init(A, lda, 0, (unsigned)((TA ? ((long)(K - 1) * lda + M) : ((long)(M - 1) * lda + K)) * 2), w, lane, 0,
          A_PIECES / 2 + w * CW);
  int cf_tile = 0, cf_k = 0;
  unsigned cf_a = 0;
  (void)cf_a;  // read in the device pass only
  auto cf_set = [&](int idx) {
    int tm = 0, tn = 0;
    if (idx < my_tiles) origin(idx, tm, tn);
    const long m0 = (long)tm * BM;
    cf_a = (unsigned)((TA ? m0 : m0 * lda) * 2);
  };
  auto cf_issue = [&](char* stage) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int sa = (int)(cf_a + (unsigned)(cf_k * BK) * (unsigned)fc.kstep);
#pragma unroll
    for (int i = 0; i < CW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(fc.rsrc, (SD_LDS void*)(stage + (A_PIECES / 2 + w * CW + i) * 1024), 16,
                                               fc.voff[i], sa, 0, 0);
#endif
    if (++cf_k == nk) { cf_k = 0; cf_set(++cf_tile); }
  };
  cf_set(0);
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) cf_issue(smem + d * STAGE);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                 // the first DEPTH K-steps have landed
  if (half == 1) __builtin_amdgcn_s_barrier();  // second half runs one phase behind

  int cur_i = 0, cnx_i = DEPTH, ck = 0, ctile = 0;
  // The finished 16 x 64 blocks of this wave go through an LDS patch of the wave's own (XOR-swizzled 16-byte chunks) so
  // that a store instruction writes whole 128-byte lines instead of 16 x 4 pieces of 32 B (-1.1 us per tile).
  // Spreading the stores over the K-steps of the next tile (all CUs reach their tile boundary together) was
  // measured: no gain.
  char* ep = smem + NST * STAGE + w * PATCH;
  auto store_rows = [&](const f32x4 (&a)[4], int gm0, int gn0) {  // a: this wave's 16 rows x 64 columns (j = 0..3)
    const int r = lane & 15, q4 = lane >> 4;
    {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)a[j][e];
        *(bf16x4*)(ep + r * 128 + (((2 * j + (q4 >> 1)) ^ (r & 7)) << 4) + (q4 & 1) * 8) = o;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int rr = hh * 8 + (lane >> 3), cc = lane & 7;
        const bf16x8 v = *(const bf16x8*)(ep + rr * 128 + ((cc ^ (rr & 7)) << 4));
        const int gmr = gm0 + rr, gn = gn0 + cc * 8;
        if (gmr < M && gn < N) *(bf16x8*)(C + (long)gmr * ldc + gn) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the patch is rewritten by the next block
    }
  };
  for (int g = 0; g < total; ++g) {
    // ---- LOAD(g): my pieces of K-step g+DEPTH, fragments of K-step g
    SD_STAMP_DUMP(g);
    SD_STAMP(g, 0);
    cf_issue(smem + cnx_i * STAGE);
    cnx_i = (cnx_i == NST - 1) ? 0 : cnx_i + 1;
    const char* cur = smem + cur_i * STAGE;
    bf16x8 af[2][MT], bfr[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if constexpr (TA) {
        load_frags_tr<BM, MT>(cur, wm * WR, kk, lane, af[kk]);
      } else {
#pragma unroll
        for (int i = 0; i < MT; ++i) af[kk][i] = load_frag<TA, BM>(cur, wm * WR + i * 16, kk, lane);
      }
      if constexpr (EPI == 3) {
        static_assert(EPI != 3 || !TB, "SwiGLU epilogue: forward (NT) only");
#pragma unroll
        for (int j = 0; j < 4; ++j)
          bfr[kk][j] = load_frag<false, BN>(cur + A_BYTES, (j >> 1) * 64 + wn * 32 + (j & 1) * 16, kk, lane);
      } else if constexpr (TB) {
        load_frags_tr<BN, 4>(cur + A_BYTES, wn * 64, kk, lane, bfr[kk]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[kk][j] = load_frag<TB, BN>(cur + A_BYTES, wn * 64 + j * 16, kk, lane);
      }
    }
    SD_STAMP(g, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * CW) : "memory");  // my pieces of K-step g+1 have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SD_STAMP(g, 2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    SD_STAMP(g, 3);
    // ---- COMPUTE(g)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(bfr[kk][j], af[kk][i], acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
    if (++ck == nk) {  // tile finished: this wave's block leaves, clear, go on with the next tile
      ck = 0;
      int tm, tn;
      origin(ctile++, tm, tn);
      const int m0 = tm * BM;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if constexpr (EPI == 3) {
          const int gm = m0 + wm * WR + i * 16 + (lane & 15);
          // folded RMSNorm: this row's rstd from the producers' table of this tile (parity of the tile index)
          const float rsc = ea.ssq_in ? ((const float*)(smem + NST * STAGE + ((ctile - 1) & 1) * 1024))[wm * WR + i * 16 + (lane & 15)]
                                      : 1.f;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int gc = tn * 64 + wn * 32 + j * 16 + (lane >> 4) * 4;  // column of act; gate at gc, up at I + gc
            bf16x4 a4, g4, u4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              // same rounding as the unfused pair: gate|up are rounded to bf16 first (sd_swiglu_fwd reads them back)
              const bf16 gb = (bf16)(acc[i][j][e] * rsc), ub = (bf16)(acc[i][j + 2][e] * rsc);
              const float gf = (float)gb, uf = (float)ub;
              g4[e] = gb; u4[e] = ub;
              a4[e] = (bf16)(gf / (1.f + __expf(-gf)) * uf);
            }
            if (gm < M && gc < ea.I) {
              *(bf16x4*)(ea.out2 + (long)gm * ea.ld2 + gc) = a4;
              if (C) {
                *(bf16x4*)(C + (long)gm * ldc + gc) = g4;
                *(bf16x4*)(C + (long)gm * ldc + ea.I + gc) = u4;
              }
            }
          }
        } else {
          store_rows(acc[i], m0 + wm * WR + i * 16, tn * BN + wn * 64);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    SD_STAMP(g, 4);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    SD_STAMP(g, 5);
    cur_i = (cur_i == NST - 1) ? 0 : cur_i + 1;
  }
  if (half == 0) __builtin_amdgcn_s_barrier();  // re-align the halves
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain my tail prefetches before the workgroup retires
}

#undef SD_STAMP
#undef SD_STAMP_DUMP

// ---------------------------------------------------------------------------------------------------
// Persistent 256 x 256 kernel for the forward (NT) GEMMs with many output columns: lm_head (N = vocabulary) and gate|up.
// Every kernel above is bound by the L2 -> LDS rate (one CU's LDS-DMA sustains ~59 GB/s), i.e. by FLOP per staged
// byte, i.e. by tile area: 256 x 128 stages (256+128)*2 B per 2*256*128 FLOP of a k (85 FLOP/B, 0.81 us per 64-deep
// K-step, 1.3 PFLOP/s in the loop); 256 x 256 stages 128 FLOP/B -- the same bytes per CU and K-step as 1.33 tiles of
// 256 x 128 for twice the FLOPs.  The price is registers: a wave owns 64 x 128 of C = 128 accumulator registers, so
// there are 8 waves (two per SIMD, 256 registers each) and no producer waves: the two 4-wave halves run the
// staggered LOAD / COMPUTE protocol of gemm_stag_kernel and issue their own LDS-DMA.  LDS cannot hold three 64-deep
// stages of a 256 x 256 tile (3 x 64 KiB), so the ring is NST = 4 stages of BK = 32 (32 KiB each, image rows of
// 64 B = 4 chunks, XOR-swizzled by swz32(row)), three K-steps in flight; a phase is still 32 MFMAs per wave.
//   LOAD(g):    issue A of K-step g+3 -> stage (g+3)%4 | ds_read K-step g | vmcnt(3*NI): my pieces of g+1 landed | lgkmcnt(0)
//   COMPUTE(g): issue B of K-step g+3 | 32 x v_mfma_f32_16x16x32_bf16
// Hazards as gemm_stag_kernel with NST = DEPTH + 1: RAW K-step x is first read in phase 2x, retired by every wave's
// counted vmcnt at the end of its LOAD(x-1) and a barrier; WAR the stage of K-step x-1 is refilled from phase 2x on,
// its last reads completed (lgkmcnt(0)) before the closing barriers of phases 2x-2 / 2x-1.
// A workgroup walks its tiles as ONE K stream (the tile origin is a scalar offset of the buffer load); a finished
// 16 x 64 block leaves through the wave's 2 KiB LDS patch as whole 128-byte lines.
//   EPI 0: C = A.B^T     EPI 3: SwiGLU, B tile = 128 gate rows | 128 up rows: gate and up of one output sit in one lane.
// PAIR (round 3): the 32-deep stages above fetch every 128-byte line of an operand row TWICE -- a stage's row is 64 B, half a
// line, and the other half is requested again one K-step later, long after the 32 KiB L1 has turned over: the PMC pass
// counts 92.2 M L1->L2 requests per lm_head launch for 46 M lines (profiles/r03_pmc_operand_stream.json), i.e. the kernel
// sat at the 16 TB/s L2->L1 ceiling with half of it wasted.  With PAIR the ring is two slots of 64 KiB, each holding
// BOTH 32-deep halves of a 64-deep K range: a 1 KiB DMA piece is 8 rows x 128 B of the operand (lanes 0-31 the first
// k-half, lanes 32-63 the second: whole lines, one request each) and lands as [8 rows x 64 B | 8 rows x 64 B]; the
// fragment reads of step g take half g & 1 of slot (g >> 1) & 1.  A slot is refilled during the EVEN step after its
// last use (A pieces in the LOAD phase, B pieces at the head of the COMPUTE phase) and waited for (vmcnt(0)) in the odd
// step's LOAD phase, one full step before its first read.  Measured (tests/bench_p256_pair.py): half the L2 requests,
// identical results, 1.7 % faster (teacher lm_head 885.8 -> 870.8 us, student 497.2 -> 488.8 us): the kernel is paced by
// its LOAD / COMPUTE phase structure, not by the L2 -> L1 rate it no longer saturates.
//   RAW  super-stage q+1 (steps 2q+2, 2q+3) is first read in phase 4q+4; every wave waits vmcnt(0) in its LOAD(2q+1)
//        (phases 4q+2 / 4q+3) and passes a barrier.
//   WAR  slot (q+1)&1 is refilled from phase 4q on; its last reads (step 2q-1) were issued in phases 4q-2 / 4q-1 and
//        completed (lgkmcnt(0)) before those phases' closing barriers.
template <int EPI, bool PAIR>
__global__ __launch_bounds__(512) void gemm_p256_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* C, int M,
                                                        int N, int K, long lda, long ldb, long ldc, int tiles_m,
                                                        int tiles_n, int group_m, EpiArgs ea) {
  static_assert(EPI == 0 || EPI == 3, "plain or SwiGLU epilogue");
  constexpr int BM = 256, BNN = 256, NW = 8, NST = 4, DEPTH = NST - 1;
  constexpr int SLOT = 2 * (BM + BNN) * P2_BK * 2, SLOT_A = 2 * BM * P2_BK * 2;  // PAIR: 64 KiB per slot, 32 KiB of it A
  constexpr int NIP = 2 * BM * P2_BK * 2 / 1024 / NW;                            // PAIR: 4 pieces of A and 4 of B per wave and slot
  constexpr int A_BYTES = BM * P2_BK * 2, STAGE = (BM + BNN) * P2_BK * 2;  // 16 / 32 KiB
  constexpr int NI = BM * P2_BK * 2 / 1024 / NW;                           // 2 pieces of A and 2 of B per wave and K-step
  constexpr int LOADS = 2 * NI;
  constexpr int PATCH = 2048;
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE + NW * PATCH];  // 144 KiB
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int wm = w >> 1, wn = w & 1, half = w >> 2;
  const int ntiles = tiles_m * tiles_n;
  const int nk = K / P2_BK;
  const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_tiles * nk;
  auto origin = [&](int idx, int& tm, int& tn) {
    const int t = xcd_remap((int)blockIdx.x + idx * (int)gridDim.x, ntiles);
    tile_coords(t, tiles_m, tiles_n, group_m, tm, tn);
  };
  // ---- LDS-DMA of this wave: loop-invariant per-lane offsets, the tile origin and k are scalar
  int voff_a[PAIR ? NIP : NI], voff_b[PAIR ? NIP : NI];
  if constexpr (PAIR) {
#pragma unroll
    for (int i = 0; i < NIP; ++i) {
      const int blk = w * NIP + i;  // 8 rows of the 256-row operand tile
      const int r = blk * 8 + ((lane & 31) >> 2), kh = lane >> 5, c = (lane & 3) ^ swz32(r);
      voff_a[i] = (int)(((long)r * lda + kh * P2_BK + c * 8) * 2);
      voff_b[i] = (int)(((long)(r + ((EPI == 3 && r >= 128) ? ea.I - 128 : 0)) * ldb + kh * P2_BK + c * 8) * 2);
    }
  } else {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int p = (w * NI + i) * 64 + lane;
      const int r = p >> 2, c = (p & 3) ^ swz32(r);
      voff_a[i] = (int)(((long)r * lda + c * 8) * 2);
      voff_b[i] = (int)(((long)(r + ((EPI == 3 && r >= 128) ? ea.I - 128 : 0)) * ldb + c * 8) * 2);
    }
  }
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(((long)(M - 1) * lda + K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(((long)(N - 1) * ldb + K) * 2), 0x00020000);
#endif
  int pf_tile = 0, pf_k = 0;
  unsigned pf_a = 0, pf_b = 0;
  (void)pf_a; (void)pf_b;  // read in the device pass only
  auto pf_set = [&](int idx) {
    int tm = 0, tn = 0;
    if (idx < my_tiles) origin(idx, tm, tn);
    pf_a = (unsigned)((long)tm * BM * lda * 2);
    pf_b = (unsigned)((long)tn * (EPI == 3 ? 128 : BNN) * ldb * 2);
  };
  // one K-step = A pieces (issued in the wave's LOAD phase) + B pieces (issued at the head of its COMPUTE phase, so that
  // a LOAD phase -- which the other half's 512 cycles of MFMA have to cover -- carries 2 DMA issues + 12 LDS reads and
  // not 4 + 12).  K-steps past the end re-read the first origin and are never used.
  auto pf_issue_a = [&](char* stage) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int sa = (int)(pf_a + (unsigned)(pf_k * P2_BK * 2));
    if constexpr (PAIR) {
#pragma unroll
      for (int i = 0; i < NIP; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (SD_LDS void*)(stage + (w * NIP + i) * 1024), 16, voff_a[i], sa, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < NI; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (SD_LDS void*)(stage + (w * NI + i) * 1024), 16, voff_a[i], sa, 0, 0);
    }
#endif
  };
  auto pf_issue_b = [&](char* stage) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int sb = (int)(pf_b + (unsigned)(pf_k * P2_BK * 2));
    if constexpr (PAIR) {
#pragma unroll
      for (int i = 0; i < NIP; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (SD_LDS void*)(stage + SLOT_A + (w * NIP + i) * 1024), 16, voff_b[i], sb, 0, 0);
      pf_k += 2;  // a slot covers two 32-deep K-steps (nk is even: K % 64 == 0)
      if (pf_k >= nk) { pf_k = 0; pf_set(++pf_tile); }
    } else {
#pragma unroll
      for (int i = 0; i < NI; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (SD_LDS void*)(stage + A_BYTES + (w * NI + i) * 1024), 16, voff_b[i], sb, 0, 0);
      if (++pf_k == nk) { pf_k = 0; pf_set(++pf_tile); }
    }
#endif
  };
  pf_set(0);
  if constexpr (PAIR) {
    pf_issue_a(smem);
    pf_issue_b(smem);
  } else {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { pf_issue_a(smem + d * STAGE); pf_issue_b(smem + d * STAGE); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (half == 1) __builtin_amdgcn_s_barrier();  // second half runs one phase behind

  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  char* ep = smem + NST * STAGE + w * PATCH;
  // a: 16 rows x 64 columns of this wave (4 fragments) -> dst rows gm0.., columns gn0.. as whole 128-byte lines
  auto store_rows = [&](const bf16x4 (&o)[4], bf16* dst, long ldd, int gm0, int gn0, int nlim) {
    const int r = lane & 15, q4 = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) *(bf16x4*)(ep + r * 128 + (((2 * j + (q4 >> 1)) ^ (r & 7)) << 4) + (q4 & 1) * 8) = o[j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int rr = hh * 8 + (lane >> 3), cc = lane & 7;
      const bf16x8 v = *(const bf16x8*)(ep + rr * 128 + ((cc ^ (rr & 7)) << 4));
      const int gmr = gm0 + rr, gn = gn0 + cc * 8;
      if (gmr < M && gn < nlim) *(bf16x8*)(dst + (long)gmr * ldd + gn) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the patch is rewritten by the next block
  };
  int cur_i = 0, nxt_i = DEPTH, ck = 0, ctile = 0;
  // PAIR: fragment (16 rows from row16_base, k-half kh) of the slot image: blocks of 8 rows, [8 x 64 B half 0 | 8 x 64 B half 1]
  auto load_frag_pair = [&](const char* opnd, int row16_base, int kh) __attribute__((always_inline)) {
    const int r = row16_base + (lane & 15), c = lane >> 4;
    return *(const bf16x8*)(opnd + (r >> 3) * 1024 + kh * 512 + (r & 7) * 64 + ((c ^ swz32(r)) << 4));
  };
  for (int g = 0; g < total; ++g) {
    // ---- LOAD(g)
    bf16x8 af[4], bfr[8];
    if constexpr (PAIR) {
      const int kh = g & 1, slot = (g >> 1) & 1;
      if (kh == 0) pf_issue_a(smem + (slot ^ 1) * SLOT);  // the other slot's last reads ended a full step ago
      const char* cur = smem + slot * SLOT;
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = load_frag_pair(cur, wm * 64 + i * 16, kh);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int brow = (EPI == 3) ? (j >> 2) * 128 + wn * 64 + (j & 3) * 16 : wn * 128 + j * 16;
        bfr[j] = load_frag_pair(cur + SLOT_A, brow, kh);
      }
      // odd step: everything this wave staged for the next slot (issued during the even step) has landed
      if (kh == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      pf_issue_a(smem + nxt_i * STAGE);
      const char* cur = smem + cur_i * STAGE;
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = load_frag32(cur, wm * 64 + i * 16, lane);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int brow = (EPI == 3) ? (j >> 2) * 128 + wn * 64 + (j & 3) * 16 : wn * 128 + j * 16;
        bfr[j] = load_frag32(cur + A_BYTES, brow, lane);
      }
      // issued so far, youngest first: A(g+3), B(g+2), A(g+2), B(g+1), ...: K-step g+1 has landed when all but 3*NI are done
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NI) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- COMPUTE(g)
    if constexpr (PAIR) {
      if ((g & 1) == 0) pf_issue_b(smem + (((g >> 1) & 1) ^ 1) * SLOT);
    } else {
      pf_issue_b(smem + nxt_i * STAGE);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = mfma16(bfr[j], af[i], acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
    if (++ck == nk) {  // tile finished: this wave's 64 x 128 block leaves, clear, go on with the next tile
      ck = 0;
      int tm, tn;
      origin(ctile++, tm, tn);
      const int m0 = tm * BM;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int gm0 = m0 + wm * 64 + i * 16;
        if constexpr (EPI == 3) {
          // act columns tn*128 + wn*64 .. +63: gate in acc[i][0..3], up in acc[i][4..7] (same lane, same output)
          bf16x4 a4[4], g4[4], u4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              // same rounding as the unfused pair: gate|up are rounded to bf16 first (sd_swiglu_fwd reads them back)
              const bf16 gb = (bf16)acc[i][j][e], ub = (bf16)acc[i][j + 4][e];
              const float gf = (float)gb, uf = (float)ub;
              g4[j][e] = gb; u4[j][e] = ub;
              a4[j][e] = (bf16)(gf / (1.f + __expf(-gf)) * uf);
            }
          const int gc0 = tn * 128 + wn * 64;
          store_rows(a4, ea.out2, ea.ld2, gm0, gc0, ea.I);
          if (C) {
            store_rows(g4, C, ldc, gm0, gc0, ea.I);
            store_rows(u4, C + ea.I, ldc, gm0, gc0, ea.I);
          }
        } else {
#pragma unroll
          for (int jh = 0; jh < 2; ++jh) {
            bf16x4 o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int e = 0; e < 4; ++e) o[j][e] = (bf16)acc[i][4 * jh + j][e];
            store_rows(o, C, ldc, gm0, tn * BNN + wn * 128 + jh * 64, N);
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    cur_i = (cur_i == NST - 1) ? 0 : cur_i + 1;
    nxt_i = (nxt_i == NST - 1) ? 0 : nxt_i + 1;
  }
  if (half == 0) __builtin_amdgcn_s_barrier();  // re-align the halves
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the tail prefetches before the workgroup retires
}

// ---------------------------------------------------------------------------------------------------
// Grouped form of the persistent kernel for the weight gradients of one decoder layer: up to 4 independent problems
// C_p[M_p,N_p] = A_p^T . B_p with a common contraction length K (the tokens of the micro-batch), both operands stored
// [K][rows] (dY and X as they lie in HBM).  Separately the four dW GEMMs of a Qwen3 layer have 64-192 tiles of
// 256x128 each, so none of them fills 256 CUs with the big tile and each pays its own launch and ramp; as one
// persistent launch their 480 tiles are one stream of work (1.9 tiles per CU).  Same 12-wave structure and hazard
// argument as gemm_pstag_kernel; a tile id is looked up in the problem table (kernel argument) by both roles.
struct GroupArgs {
  const bf16* A[4];
  const bf16* B[4];
  bf16* C[4];
  long lda[4], ldb[4], ldc[4];
  int M[4], N[4], tiles_m[4], tiles_n[4];
  int start[5];  // first global tile id of each problem; start[n] = total
  int n;
};

// SHARE: the compute waves issue 2 of the 48 LDS-DMA pieces of a stage each (the second half of A), as in gemm_pstag_kernel
template <bool ACCUM, bool SHARE>
__global__ __launch_bounds__(768) void gemm_pgroup_tn_kernel(GroupArgs ga, int K, int group_m) {
  constexpr int BM = 256, NW = 8, NPROD = 4, NST = 3, DEPTH = NST - 1;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;  // 48 KiB
  constexpr int A_PIECES = BM / 8, B_PIECES = BN / 8;
  constexpr int CW = (A_PIECES / 2) / NW;
  constexpr int PA = (SHARE ? A_PIECES / 2 : A_PIECES) / NPROD, PB = B_PIECES / NPROD;
  constexpr int LOADS = PA + PB;
  constexpr int PATCH = 2048;
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE + NW * PATCH];
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int ntiles = ga.start[ga.n];
  const int nk = (K + BK - 1) / BK;
  const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_tiles * nk;
  // (problem, tile row, tile column) of my idx-th tile
  auto locate = [&](int idx, int& p, int& tm, int& tn) {
    const int t = xcd_remap((int)blockIdx.x + idx * (int)gridDim.x, ntiles);
    p = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q)
      if (q < ga.n && t >= ga.start[q]) p = q;
    p = __builtin_amdgcn_readfirstlane(p);
    const int gm = group_m < ga.tiles_m[p] ? group_m : ga.tiles_m[p];
    tile_coords(t - ga.start[p], ga.tiles_m[p], ga.tiles_n[p], gm, tm, tn);
  };

  if (w >= NW) {  // ------------------------------------------------------------ producer waves
    const int pw = w - NW;
    FastStage<true, BM, A_PIECES / PA> fa;  // pieces pw*PA .. +PA of the A tile (its first half when the compute waves share)
    FastStage<true, BN, NPROD> fb;
    static_assert(FastStage<true, BM, A_PIECES / PA>::NI == PA && FastStage<true, BN, NPROD>::NI == PB, "piece split");
    int pf_tile = 0, pf_k = 0;
    unsigned pf_a = 0, pf_b = 0;
    (void)pf_a; (void)pf_b;
    auto pf_set = [&](int idx) {
      int p = 0, tm = 0, tn = 0;
      if (idx < my_tiles) locate(idx, p, tm, tn);
      fa.init(ga.A[p], ga.lda[p], 0, (unsigned)(((long)(K - 1) * ga.lda[p] + ga.M[p]) * 2), pw, lane, 0, pw * PA);
      fb.init(ga.B[p], ga.ldb[p], 0, (unsigned)(((long)(K - 1) * ga.ldb[p] + ga.N[p]) * 2), pw, lane);
      pf_a = (unsigned)((long)tm * BM * 2);
      pf_b = (unsigned)((long)tn * BN * 2);
    };
    auto pf_issue = [&](char* stage) {
#if defined(__HIP_DEVICE_COMPILE__)
      const int sa = (int)(pf_a + (unsigned)(pf_k * BK) * (unsigned)fa.kstep);
      const int sb = (int)(pf_b + (unsigned)(pf_k * BK) * (unsigned)fb.kstep);
#pragma unroll
      for (int i = 0; i < PA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(fa.rsrc, (SD_LDS void*)(stage + (pw * PA + i) * 1024), 16, fa.voff[i], sa, 0, 0);
#pragma unroll
      for (int i = 0; i < FastStage<true, BN, NPROD>::NI; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            fb.rsrc, (SD_LDS void*)(stage + A_BYTES + (pw * FastStage<true, BN, NPROD>::NI + i) * 1024), 16, fb.voff[i],
            sb, 0, 0);
#endif
      if (++pf_k == nk) { pf_k = 0; pf_set(++pf_tile); }
    };
    pf_set(0);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) pf_issue(smem + d * STAGE);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int nxt = DEPTH;
    for (int g = 0; g < total; ++g) {
      pf_issue(smem + nxt * STAGE);
      nxt = (nxt == NST - 1) ? 0 : nxt + 1;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * LOADS) : "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ---------------------------------------------------------------------------- compute waves
  const int wm = w >> 1, wn = w & 1;
  const int half = w >> 2;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  FastStage<true, BM, A_PIECES / CW> fc;
  int cf_tile = 0, cf_k = 0, cnx_i = DEPTH;
  unsigned cf_a = 0;
  (void)cf_a; (void)cf_tile; (void)cf_k; (void)cnx_i;
  auto cf_set = [&](int idx) {
    int p = 0, tm = 0, tn = 0;
    if (idx < my_tiles) locate(idx, p, tm, tn);
    fc.init(ga.A[p], ga.lda[p], 0, (unsigned)(((long)(K - 1) * ga.lda[p] + ga.M[p]) * 2), w, lane, 0, A_PIECES / 2 + w * CW);
    cf_a = (unsigned)((long)tm * BM * 2);
  };
  auto cf_issue = [&](char* stage) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int sa = (int)(cf_a + (unsigned)(cf_k * BK) * (unsigned)fc.kstep);
#pragma unroll
    for (int i = 0; i < CW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(fc.rsrc, (SD_LDS void*)(stage + (A_PIECES / 2 + w * CW + i) * 1024), 16,
                                               fc.voff[i], sa, 0, 0);
#endif
    if (++cf_k == nk) { cf_k = 0; cf_set(++cf_tile); }
  };
  if constexpr (SHARE) {
    cf_set(0);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) cf_issue(smem + d * STAGE);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (half == 1) __builtin_amdgcn_s_barrier();

  int cur_i = 0, ck = 0, ctile = 0;
  char* ep = smem + NST * STAGE + w * PATCH;
  for (int g = 0; g < total; ++g) {
    if constexpr (SHARE) {
      cf_issue(smem + cnx_i * STAGE);
      cnx_i = (cnx_i == NST - 1) ? 0 : cnx_i + 1;
    }
    const char* cur = smem + cur_i * STAGE;
    bf16x8 af[2][4], bfr[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      load_frags_tr<BM, 4>(cur, wm * 64, kk, lane, af[kk]);
      load_frags_tr<BN, 4>(cur + A_BYTES, wn * 64, kk, lane, bfr[kk]);
    }
    if constexpr (SHARE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * CW) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(bfr[kk][j], af[kk][i], acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
    if (++ck == nk) {
      ck = 0;
      int p, tm, tn;
      locate(ctile++, p, tm, tn);
      bf16* C = ga.C[p];
      const long ldc = ga.ldc[p];
      const int M = ga.M[p], N = ga.N[p];
      const int r = lane & 15, q4 = lane >> 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int gm0 = tm * BM + wm * 64 + i * 16, gn0 = tn * BN + wn * 64;
        bf16x4 prev[4];
        if constexpr (ACCUM) {  // C += ...: the old value joins the fp32 sum before the one rounding (as sd_gemm_bf16 with R = C)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int gmr = gm0 + r, gn = gn0 + j * 16 + q4 * 4;
            prev[j] = (gmr < M && gn < N) ? *(const bf16x4*)(C + (long)gmr * ldc + gn) : bf16x4{};
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16)(ACCUM ? acc[i][j][e] + (float)prev[j][e] : acc[i][j][e]);
          *(bf16x4*)(ep + r * 128 + (((2 * j + (q4 >> 1)) ^ (r & 7)) << 4) + (q4 & 1) * 8) = o;
          acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int rr = hh * 8 + (lane >> 3), cc = lane & 7;
          const bf16x8 v = *(const bf16x8*)(ep + rr * 128 + ((cc ^ (rr & 7)) << 4));
          const int gmr = gm0 + rr, gn = gn0 + cc * 8;
          if (gmr < M && gn < N) *(bf16x8*)(C + (long)gmr * ldc + gn) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    cur_i = (cur_i == NST - 1) ? 0 : cur_i + 1;
  }
  if (half == 0) __builtin_amdgcn_s_barrier();
  if constexpr (SHARE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// C = sum_s slab[s] (+ R), fixed order
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, bf16* C, const bf16* R, int M,
                                                            int N, long ldc, long ldr, int splits) {
  const long n8 = (long)M * N / 8;
  const int row8 = N / 8;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n8; q += (long)gridDim.x * 256) {
    const long m = q / row8;
    const int n = (int)(q % row8) * 8;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < splits; ++s) {
      const float* p = slabs + ((long)s * M + m) * N + n;
      f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
      v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3]; v[4] += b[0]; v[5] += b[1]; v[6] += b[2]; v[7] += b[3];
    }
    if (R) {
      bf16x8 r = *(const bf16x8*)(R + m * ldr + n);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
    *(bf16x8*)(C + m * ldc + n) = o;
  }
}

#ifdef SD_STAMPS
void* g_stamp_buffer = nullptr;  // diagnostic build: device buffer for the phase stamps of gemm_pstag_kernel
#endif
// Measurement switches live in g_sd_debug (sd_debug.h, set through include/sd_hip_debug.h); defaults = product behaviour.
// gemm_cu_budget (multi-GPU runs): the persistent kernels of the BACKWARD (grouped weight gradients, lm_head weight
// gradient) take one workgroup per CU for their whole duration; when RCCL's reduction kernels hold c CUs meanwhile, c
// workgroups start only after others have finished and the launch takes up to twice as long (measured with
// bench.py --experiment-cu-hog 16: +22 % / +70 %).  A budget of 256 - c keeps every workgroup resident from the start.
// Default (knob 0): when the backward runs them on its side stream beside the dX chain (t_sd_shared_gpu, set by the
// runner), three quarters of the CUs -- the chain's kernels then always find CUs instead of queueing behind a launch
// that holds every one of them for 70-550 us: 19.51 -> 19.31 ms per config-2 step and half the run-to-run spread
// (tests/bench_knob_ab.py gemm.cu_budget 0 192, alternating in one process; 224 / 160 / 128 measured no better).
// Only for 1 536 .. 3 072 tokens per launch (K of a weight-gradient GEMM): measured -0.16 / -0.20 / -0.19 ms at 1 536 /
// 2 048 / 3 072 tokens, +-0.0 at 1 024, +0.1 at 512, and +0.57 / +1.3 ms at 4 096 / 8 192, where the chain's own kernels
// fill the chip for long stretches and the budget only slows the weight gradients.  Knob -1: one workgroup per CU always.
static int cu_budget(int cus, int tokens) {
  if (g_sd_debug.gemm_cu_budget > 0) return g_sd_debug.gemm_cu_budget & ~7;
  const bool in_range = tokens >= 1536 && tokens <= 3072;
  return (g_sd_debug.gemm_cu_budget == 0 && t_sd_shared_gpu && in_range) ? ((cus * 3 / 4) & ~7) : 0;
}
thread_local bool g_skip_reduce = false;  // set by sd_gemm_bf16_splitk_partial around its dispatch

template <int BM, int NST, bool TA, bool TB>
int launch(const void* A, const void* B, void* C, const void* R, float* slabs, int splits, int M, int N, int K, long lda,
           long ldb, long ldc, long ldr, int epi_kind, const EpiArgs& ea, hipStream_t st, int tflags) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int kt_all = (K + BK - 1) / BK;
  const int per = (kt_all + splits - 1) / splits;
  dim3 grid(tiles_m * tiles_n, splits), block(BM == 256 ? 512 : 256);
  // descriptor-based staging needs every k >= K to read as zero in at least one operand (transposed
  // operands get that from the hardware range check; two K-contiguous ones need K % 64 == 0) and 31-bit offsets
  const long bytes_a = (TA ? ((long)(K - 1) * lda + M) : ((long)(M - 1) * lda + K)) * 2;
  const long bytes_b = (TB ? ((long)(K - 1) * ldb + N) : ((long)(N - 1) * ldb + K)) * 2;
  const long span = ((long)kt_all + 6) * BK * 2 * (TA ? lda : 1) + bytes_a;
  const long span_b = ((long)kt_all + 6) * BK * 2 * (TB ? ldb : 1) + bytes_b;
  // persistent kernel: one workgroup per CU (a multiple of 8 so that every workgroup stays on its XCD's tile run)
  static const int all_cus = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return 0;
    return cus & ~7;
  }();
  // tflags: what the measured table (sd_gemm_table.inc) asks for this very shape: 1 no persistent kernel, 2 no 256x256
  // kernel, 4 the 256x256 kernel whatever the tile count
  const int persist_grid = (g_sd_debug.gemm_no_persist || (tflags & 1)) ? 0 : all_cus;
  const bool p256_ok = !g_sd_debug.gemm_no_p256 && !(tflags & 2);
  const bool p256_pair = !g_sd_debug.gemm_p256_unpaired;
  // Measured (tests/bench_p256.py, MI355X): the 256 x 256 kernel ties the 256 x 128 one on the lm_head class (544 vs
  // 557 us student, 924 vs 929 us teacher) and loses on gate|up (114 vs 93 us teacher, 37.6 vs 36.3 us student): both
  // settle at ~0.9 us per staged K-step whatever the bytes of the step, i.e. the loop is paced by the latency of the
  // operand stream at the LDS-limited prefetch depth, not by L2 -> LDS bandwidth per FLOP.  So only the vocabulary-wide
  // GEMMs take it.  gemm.p256_min_tiles (sd_hip_debug.h) lowers the threshold for measurements / tests.
  const int p256_min_tiles = (tflags & 4) ? 1 : g_sd_debug.gemm_p256_min_tiles;
  const int gm_env = g_sd_debug.gemm_group_m;
  int gm = gm_env > 0 ? gm_env : (BM == 256 ? 4 : 8);
  if (gm > tiles_m) gm = tiles_m;
  const bool fast = !g_sd_debug.gemm_checked_staging && (TA || TB || (K % BK) == 0) && span < 0x7fffffffL && span_b < 0x7fffffffL;
#ifdef SD_STAMPS
#define SD_STAMP_ARGS() EpiArgs ea_st = ea; ea_st.cos_t = (const bf16*)g_stamp_buffer
#define SD_STAMP_EA ea_st
#else
#define SD_STAMP_ARGS() do { } while (0)
#define SD_STAMP_EA ea
#endif
#define SD_GEMM_GO(EPI)                                                                                              \
  do {                                                                                                               \
    if constexpr (BM == 256 && NST == 9 && !TA && !TB && (EPI == 0 || EPI == 3)) {                                   \
      /* many-column forward GEMMs: 256 x 256 tiles when they still fill >= 70 % of the CUs' rounds */               \
      const int t_m = (M + 255) / 256, t_n = (EPI == 3) ? (ea.I + 127) / 128 : (N + 255) / 256, nt2 = t_m * t_n;       \
      const int rounds = persist_grid > 0 ? (nt2 + persist_grid - 1) / persist_grid : 0;                              \
      if (splits == 1 && !R && !ea.ssq_in && p256_ok && persist_grid > 0 && (K % P2_BK) == 0 &&                         \
          (EPI != 3 || (ea.I % 128) == 0) &&                                                                          \
          (N % 8) == 0 && nt2 >= p256_min_tiles && 10 * nt2 >= 7 * rounds * persist_grid && span < 0x7fffffffL &&      \
          span_b < 0x7fffffffL) {                                                                                     \
        const int grid2 = nt2 > persist_grid ? persist_grid : nt2;                                                    \
        if (p256_pair && (K % (2 * P2_BK)) == 0) {                                                                    \
          SD_PROF_LABEL("gemm_p256_kernel<%d, true>", EPI);                                                           \
          hipLaunchKernelGGL((gemm_p256_kernel<EPI, true>), dim3(grid2), dim3(512), 0, st, (const bf16*)A,             \
                             (const bf16*)B, (bf16*)C, M, N, K, lda, ldb, ldc, t_m, t_n, t_m < 8 ? t_m : 8, ea);        \
        } else {                                                                                                       \
          SD_PROF_LABEL("gemm_p256_kernel<%d, false>", EPI);                                                          \
          hipLaunchKernelGGL((gemm_p256_kernel<EPI, false>), dim3(grid2), dim3(512), 0, st, (const bf16*)A,            \
                             (const bf16*)B, (bf16*)C, M, N, K, lda, ldb, ldc, t_m, t_n, t_m < 8 ? t_m : 8, ea);        \
        }                                                                                                              \
        break;                                                                                                         \
      }                                                                                                                \
    }                                                                                                                  \
    if constexpr (BM == 256 && NST == 9 && (EPI == 0 || EPI == 3)) {                                                 \
      if (splits == 1 && tiles_m * tiles_n > persist_grid && persist_grid > 0) {                                       \
        SD_PROF_LABEL("gemm_pstag_kernel<4, %s, %s, %d>", TA ? "true" : "false", TB ? "true" : "false", EPI);          \
        SD_STAMP_ARGS();                                                                                               \
        /* weight gradients (TA): the backward's persistent launches honour the CU budget of a multi-GPU run */       \
        const int cb = cu_budget(persist_grid, K);                                                                     \
        int pg = (TA && cb > 0 && cb < persist_grid) ? cb : persist_grid;                                              \
        if (!TA && g_sd_debug.gemm_fwd_cu_budget > 0 && t_sd_shared_gpu && g_sd_debug.gemm_fwd_cu_budget < pg)         \
          pg = g_sd_debug.gemm_fwd_cu_budget & ~7; /* (measurement) forward persistent launches beside another stream */ \
        if (!TA && g_sd_debug.gemm_persist_balance) { /* (measurement) equal tiles per workgroup */                    \
          const int nt = tiles_m * tiles_n, rounds = (nt + persist_grid - 1) / persist_grid;                           \
          pg = (((nt + rounds - 1) / rounds) + 7) & ~7;                                                                \
          if (pg > persist_grid) pg = persist_grid;                                                                    \
        }                                                                                                              \
        hipLaunchKernelGGL((gemm_pstag_kernel<4, TA, TB, EPI>), dim3(pg), dim3(768), 0, st, (const bf16*)A,            \
                           (const bf16*)B, (bf16*)C, (const bf16*)R, M, N, K, lda, ldb, ldc, ldr, tiles_m, tiles_n,    \
                           gm, SD_STAMP_EA);                                                                           \
        break;                                                                                                         \
      }                                                                                                                \
    }                                                                                                                  \
    if constexpr (BM == 256 && NST == 9) {                                                                           \
      SD_PROF_LABEL("gemm_stag_kernel<%s, %s, %d>", TA ? "true" : "false", TB ? "true" : "false", EPI);                \
      hipLaunchKernelGGL((gemm_stag_kernel<TA, TB, EPI>), grid, block, 0, st, (const bf16*)A, (const bf16*)B,          \
                         (bf16*)C, (const bf16*)R, slabs, M, N, K, lda, ldb, ldc, ldr, tiles_m, tiles_n, per, gm, ea); \
    } else if (fast || EPI >= 3) {                                                                                   \
      SD_PROF_LABEL("gemm_bf16_kernel<%d, %d, %s, %s, %d, true>", BM, (NST == 9 ? 3 : NST), TA ? "true" : "false",     \
                    TB ? "true" : "false", EPI);                                                                       \
      hipLaunchKernelGGL((gemm_bf16_kernel<BM, (NST == 9 ? 3 : NST), TA, TB, EPI, true>), grid, block, 0, st,          \
                         (const bf16*)A, (const bf16*)B, (bf16*)C, (const bf16*)R, slabs, M, N, K, lda, ldb, ldc, ldr, \
                         tiles_m, tiles_n, per, gm, ea);                                                             \
    } else                                                                                                           \
      hipLaunchKernelGGL((gemm_bf16_kernel<BM, (NST == 9 ? 3 : NST), TA, TB, (EPI >= 3 ? 0 : EPI), false>), grid,      \
                         block, 0, st, (const bf16*)A, (const bf16*)B, (bf16*)C, (const bf16*)R, slabs, M, N, K, lda,  \
                         ldb, ldc, ldr, tiles_m, tiles_n, per, gm, ea);                                              \
  } while (0)
  if constexpr (!TA && !TB) {
    if (epi_kind == 3) { if (!fast) return SD_ERR_UNSUPPORTED; SD_GEMM_GO(3); SD_CHECK_LAUNCH(); return 0; }
    if (epi_kind == 4) { if (!fast) return SD_ERR_UNSUPPORTED; SD_GEMM_GO(4); SD_CHECK_LAUNCH(); return 0; }
  }
  if constexpr (!TA && TB) {
    if (epi_kind == 5) { if (!fast) return SD_ERR_UNSUPPORTED; SD_GEMM_GO(5); SD_CHECK_LAUNCH(); return 0; }
    if (epi_kind == 6) { if (!fast || splits != 1) return SD_ERR_UNSUPPORTED; SD_GEMM_GO(6); SD_CHECK_LAUNCH(); return 0; }
  }
  if (splits > 1) SD_GEMM_GO(2);
  else if (R) SD_GEMM_GO(1);
  else SD_GEMM_GO(0);
#undef SD_GEMM_GO
  SD_CHECK_LAUNCH();
  if (splits > 1 && !g_skip_reduce) {
    const long n8 = (long)M * N / 8;
    const int nb = (int)((n8 + 255) / 256 < 2048 ? (n8 + 255) / 256 : 2048);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nb), dim3(256), 0, st, (const float*)slabs, (bf16*)C, (const bf16*)R, M,
                       N, ldc, ldr, splits);
    SD_CHECK_LAUNCH();
  }
  return 0;
}

struct GemmTableEntry { int ta, tb, epi, M, N, K, bm, nst, flags; };
const GemmTableEntry kGemmTable[] = {
#include "sd_gemm_table.inc"
};

int check_args(const void* A, const void* B, const void* C, const void* R, int M, int N, int K, int64_t lda, int64_t ldb,
               int64_t ldc, int64_t ldr, int trans_a, int trans_b) {
  if (M <= 0 || N <= 0 || K <= 0) return SD_ERR_SHAPE;
  if ((lda | ldb | ldc | (R ? ldr : 0)) & 7) return SD_ERR_ALIGN;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)R) & 15) return SD_ERR_ALIGN;
  if (N & 7) return SD_ERR_ALIGN;
  if ((!trans_a || !trans_b) && (K & 7)) return SD_ERR_ALIGN;
  if (trans_a && (M & 7)) return SD_ERR_ALIGN;
  return 0;
}


int dispatch(const void* A, const void* B, void* C, const void* R, float* slabs, int splits, int M, int N, int K,
             long lda, long ldb, long ldc, long ldr, int ta, int tb, hipStream_t st, int epi_kind = 0,
             const EpiArgs* eap = nullptr) {
  const EpiArgs ea = eap ? *eap : EpiArgs{};
  // Picked from tests/bench_shapes.py --tune --cold on MI355X (weight operand HBM-cold, as in the real step
  // where every layer streams its own weights).  NT/NN read weights: latency-bound on HBM misses, so large
  // tiles / deeper rings win whenever the grid still covers the 256 CUs; TN (dW) reads two warm activations.
  const long tiles256 = (long)((M + 255) / 256) * ((N + BN - 1) / BN) * splits;
  const long tiles128 = (long)((M + 127) / 128) * ((N + BN - 1) / BN) * splits;
  const long blocks64 = (long)((M + 63) / 64) * ((N + BN - 1) / BN) * splits;
  int bm, nst;
  if (!ta && !tb) {  // forward linear
    if (tiles256 >= 256) { bm = 256; nst = 9; }  // 9 = the staggered two-half kernel (gemm_stag_kernel)
    else if (tiles128 >= 256) { bm = 128; nst = 3; }
    else { bm = 64; nst = blocks64 <= 320 ? 4 : 3; }
  } else if (!ta) {  // NN: dX = dY . W (weights cold)
    // round 4 (tests/bench_tune.py, profiles/r04_gemm_tune.json): the dX GEMMs with a heavy fused epilogue (EPI 5: SwiGLU
    // backward, 50 MB of gate|up / d(gate|up) per launch; EPI 6: delta) and at least 1.5 rounds of 128x128 tiles run 3-6 %
    // faster on TWO co-resident 128x128 workgroups per CU (2-stage ring, 64 KiB each: one's epilogue beside the other's K
    // loop) than on one 256x128 workgroup: down dX 29.6 -> 28.7 us at 2 048 tokens, 86.8 -> 83.2 / 53.1 -> 50.2 us at 8 192.
    if (splits == 1 && (epi_kind == 5 || epi_kind == 6) && tiles128 >= 384) { bm = 128; nst = 2; }
    else if (splits > 1 || tiles256 >= 192) { bm = 256; nst = 9; }
    else if (tiles128 >= 256) { bm = 128; nst = 3; }
    else { bm = 64; nst = 3; }
  } else {  // TN: dW = dY^T . X (two warm activations)
    const long tn_stag_min = g_sd_debug.gemm_tn_stag_min;
    if (tiles256 >= tn_stag_min) { bm = 256; nst = 9; }
    else if (tiles128 >= 320) { bm = 128; nst = 2; }
    else { bm = 64; nst = blocks64 <= 320 ? 3 : 2; }
  }
  // Dispatch by measurement (round 4): the GEMM calls of the distillation step at the BASELINE config 2 / 4 / 5 shapes were
  // timed under every variant (tests/bench_tune.py -> profiles/r04_gemm_tune.json -> scripts/make_gemm_table.py); where a
  // variant beat the heuristic above by >= 2 % the table names it.  Any other shape keeps the heuristic.
  int tflags = 0;
  if (!g_sd_debug.gemm_no_table && !g_sd_debug.gemm_force_bm) {
    const int key_epi = epi_kind ? epi_kind : (slabs ? 2 : (R ? 1 : 0));
    for (const GemmTableEntry& e : kGemmTable)
      if (e.M == M && e.N == N && e.K == K && e.ta == ta && e.tb == tb && e.epi == key_epi) {
        if (e.bm) { bm = e.bm; nst = e.nst; tflags = e.flags; }
        break;
      }
  }
  if (g_sd_debug.gemm_force_bm) { bm = g_sd_debug.gemm_force_bm; nst = g_sd_debug.gemm_force_nst; }
  // A pass that shares the GPU with a second stream (SD_FWD_CONCURRENT: the frozen teacher beside the student's forward)
  // is not served by tiles chosen to cover all 256 CUs: what counts then is CU-time per FLOP, and the other stream fills
  // whatever this launch leaves free.  The forward GEMMs that end in a residual add (o / down projections: N = hidden,
  // 256 tiles of 64 x 128 for the student, of 128 x 128 for the teacher) take the next tile up -- half as many workgroups
  // at 1.5x / 1.33x the FLOPs per staged byte: 20.46 -> 19.99 ms per config-2 step, alternating in one process
  // (tests/bench_knob_ab.py); 64 -> 256 rows and the dX GEMMs measured neutral or worse.  Only while
  // the tile that covers the chip still yields >= 192 workgroups (>= 96 after the step up): -0.46 / -0.50 / -0.85 / -0.32 ms
  // per step at 1 536 / 2 048 / 3 072 / 4 096 tokens, +-0.0 at 1 024 (128 -> 64 workgroups), +0.35 ms at 512 (64 -> 32).
  const long tiles_now = (long)((M + bm - 1) / bm) * ((N + BN - 1) / BN);
  const int bump = g_sd_debug.gemm_fwd_bump > 0 ? g_sd_debug.gemm_fwd_bump
                   : (g_sd_debug.gemm_fwd_bump == 0 && t_sd_shared_gpu && tiles_now >= 192 ? 3 : 0);
  if (bump && !ta && !tb && R && !slabs && epi_kind <= 1) {
    if (bm == 64 && (bump & 16)) { bm = 256; nst = 9; }
    else if (bm == 64 && (bump & 1)) { bm = 128; nst = (bump & 32) ? 2 : 3; }
    else if (bm == 128 && (bump & 2)) { bm = 256; nst = 9; }
    else if (bm == 128 && (bump & 64)) { nst = 2; }   // (measurement) 64 KiB of LDS: two workgroups per CU
  }
  if ((bump & 12) && !ta && tb && !slabs && splits == 1) {    // (measurement only) the same for the dX GEMMs
    if (bm == 64 && (bump & 4)) { bm = 128; nst = 3; }
    else if (bm == 128 && (bump & 8)) { bm = 256; nst = 9; }
  }
  // the staggered kernel only has the descriptor staging path
  bool stag_ok = !g_sd_debug.gemm_checked_staging && (ta || tb || (K % BK) == 0) &&
                       ((long)K * (ta ? lda : 1) + (long)M * (ta ? 1 : lda)) * 2 < 0x70000000L &&
                       ((long)K * (tb ? ldb : 1) + (long)N * (tb ? 1 : ldb)) * 2 < 0x70000000L;
  if (nst == 9 && (!stag_ok || bm != 256)) nst = 3;
  SdProfScope prof(ta ? SD_K_GEMM_TN : (tb ? SD_K_GEMM_NN : ((bm == 256 && nst == 9) ? SD_K_GEMM_NT_STAG : SD_K_GEMM_NT)),
                   2.0 * M * N * K, st);
#define SD_GO(BM_, NST_, TA_, TB_) \
  return launch<BM_, NST_, TA_, TB_>(A, B, C, R, slabs, splits, M, N, K, lda, ldb, ldc, ldr, epi_kind, ea, st, tflags)
#define SD_PICK(TA_, TB_)                                   \
  do {                                                      \
    if (bm == 256 && nst == 9 && stag_ok) SD_GO(256, 9, TA_, TB_); \
    if (bm == 256 && nst == 2) SD_GO(256, 2, TA_, TB_);     \
    if (bm == 256) SD_GO(256, 3, TA_, TB_);                 \
    if (bm == 64 && nst == 2) SD_GO(64, 2, TA_, TB_);       \
    if (bm == 64 && nst == 3) SD_GO(64, 3, TA_, TB_);       \
    if (bm == 64) SD_GO(64, 4, TA_, TB_);                   \
    if (nst == 2) SD_GO(128, 2, TA_, TB_);                  \
    if (nst == 4) SD_GO(128, 4, TA_, TB_);                  \
    SD_GO(128, 3, TA_, TB_);                                \
  } while (0)
  if (!ta && !tb) SD_PICK(false, false);
  if (!ta && tb) SD_PICK(false, true);
  if (ta && tb) SD_PICK(true, true);
  SD_PICK(true, false);
#undef SD_PICK
#undef SD_GO
}

}  // namespace

extern "C" int sd_gemm_bf16(const void* A, const void* B, void* C, const void* R, int M, int N, int K, int64_t lda,
                            int64_t ldb, int64_t ldc, int64_t ldr, int trans_a, int trans_b, void* stream) {
  if (int e = check_args(A, B, C, R, M, N, K, lda, ldb, ldc, ldr, trans_a, trans_b)) return e;
  return dispatch(A, B, C, R, nullptr, 1, M, N, K, lda, ldb, ldc, ldr, trans_a, trans_b, (hipStream_t)stream);
}

#ifdef SD_STAMPS
extern "C" void sd_debug_stamp_buffer(void* p) { g_stamp_buffer = p; }
#endif

extern "C" int sd_gemm_splitk_plan(int M, int N, int K) {
  // The kernels are bound by the L2 -> LDS rate, i.e. by FLOP per staged byte, i.e. by tile area: a GEMM whose
  // output has fewer than 256 tiles of 256x128 is split along K so that it can still use those tiles.  Measured
  // (tests/bench_shapes.py --tune --cold, whole step): worth it from K = 6144 (4 slices: gate|up dX, +1 % step) and
  // for the lm_head-class contraction (K = vocabulary, 8 slices); K <= 4096 is best unsplit on 64x128 tiles.
  const long tiles = (long)((M + 255) / 256) * ((N + BN - 1) / BN);
  const int kt = (K + BK - 1) / BK;
  const int min_kt = g_sd_debug.gemm_splitk_min_kt;
  if (tiles >= 256 || kt < min_kt) return 1;
  // The slice count that fills whole rounds of 256 workgroups best (fewest slices on a tie: every slice is another
  // fp32 slab; a slice keeps at least 24 K-steps).  48 tiles (lm_head dX on R = 1536 rows) -> 5 slices = 240
  // workgroups, not 8 = 384 = 1.5 rounds; 64 tiles -> 4.
  const int min_slice = g_sd_debug.gemm_splitk_min_slice < 1 ? 1 : g_sd_debug.gemm_splitk_min_slice;
  const int cap = g_sd_debug.gemm_splitk_max > 0 ? g_sd_debug.gemm_splitk_max : 8;  // (measurement knob)
  const int cmax = kt / min_slice < cap ? kt / min_slice : cap;
  int s = 1;
  double best = (double)tiles / 256.0;
  for (int c = 2; c <= cmax; ++c) {
    const long wg = tiles * c;
    const double eff = (double)wg / (double)(((wg + 255) / 256) * 256);
    if (eff > best + 1e-9) { best = eff; s = c; }
  }
  return s < 1 ? 1 : s;
}

extern "C" int64_t sd_gemm_splitk_workspace_bytes(int M, int N, int K) {
  const int s = sd_gemm_splitk_plan(M, N, K);
  return s > 1 ? (int64_t)s * M * N * 4 : 0;
}

extern "C" int sd_gemm_bf16_splitk(const void* A, const void* B, void* C, const void* R, int M, int N, int K, int64_t lda,
                                   int64_t ldb, int64_t ldc, int64_t ldr, int trans_a, int trans_b, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
  if (int e = check_args(A, B, C, R, M, N, K, lda, ldb, ldc, ldr, trans_a, trans_b)) return e;
  int s = sd_gemm_splitk_plan(M, N, K);
  if (s > 1 && (workspace == nullptr || workspace_bytes < (int64_t)s * M * N * 4)) s = 1;
  if (s > 1 && ((uintptr_t)workspace & 15)) return SD_ERR_ALIGN;
  return dispatch(A, B, C, R, (float*)workspace, s, M, N, K, lda, ldb, ldc, ldr, trans_a, trans_b, (hipStream_t)stream);
}

// Split-K GEMM that leaves its `*nsplit_out` fp32 slabs [nsplit][M][N] un-reduced in `workspace` for a consumer that
// sums them itself (sd_rmsnorm_bwd_slabs).  When the plan is a single slice it writes bf16 C as usual and reports 1.
extern "C" int sd_gemm_bf16_splitk_partial(const void* A, const void* B, void* C, int M, int N, int K, int64_t lda,
                                           int64_t ldb, int64_t ldc, int trans_a, int trans_b, void* workspace,
                                           int64_t workspace_bytes, int* nsplit_out, void* stream) {
  if (int e = check_args(A, B, C, nullptr, M, N, K, lda, ldb, ldc, 0, trans_a, trans_b)) return e;
  int s = sd_gemm_splitk_plan(M, N, K);
  if (s > 1 && (workspace == nullptr || workspace_bytes < (int64_t)s * M * N * 4)) s = 1;
  if (s > 1 && ((uintptr_t)workspace & 15)) return SD_ERR_ALIGN;
  *nsplit_out = s;
  g_skip_reduce = true;
  const int rc = dispatch(A, B, C, nullptr, (float*)workspace, s, M, N, K, lda, ldb, ldc, 0, trans_a, trans_b,
                          (hipStream_t)stream);
  g_skip_reduce = false;
  return rc;
}

// Weight gradients of one layer in one persistent launch: C_p [M_p,N_p] = A_p^T . B_p for p < n <= 4, A_p [K,M_p]
// (row stride lda), B_p [K,N_p], common K; accumulate: C_p += ....  SD_ERR_UNSUPPORTED when a problem does not fit the
// descriptor staging.
extern "C" int sd_gemm_grouped_tn(const sd_gemm_problem* probs, int n, int K, int accumulate, void* stream) {
  if (n <= 0 || n > 4 || K <= 0 || !probs) return SD_ERR_SHAPE;
  GroupArgs ga{};
  int start = 0;
  double flops = 0.0;
  for (int p = 0; p < n; ++p) {
    const sd_gemm_problem& q = probs[p];
    if (q.M <= 0 || q.N <= 0 || (q.M & 7) || (q.N & 7) || ((q.lda | q.ldb | q.ldc) & 7)) return SD_ERR_ALIGN;
    if (((uintptr_t)q.A | (uintptr_t)q.B | (uintptr_t)q.C) & 15) return SD_ERR_ALIGN;
    if (((long)K * q.lda + q.M) * 2 >= 0x70000000L || ((long)K * q.ldb + q.N) * 2 >= 0x70000000L) return SD_ERR_UNSUPPORTED;
    ga.A[p] = (const bf16*)q.A; ga.B[p] = (const bf16*)q.B; ga.C[p] = (bf16*)q.C;
    ga.lda[p] = q.lda; ga.ldb[p] = q.ldb; ga.ldc[p] = q.ldc;
    ga.M[p] = q.M; ga.N[p] = q.N;
    ga.tiles_m[p] = (q.M + 255) / 256; ga.tiles_n[p] = (q.N + BN - 1) / BN;
    ga.start[p] = start;
    start += ga.tiles_m[p] * ga.tiles_n[p];
    flops += 2.0 * q.M * q.N * K;
  }
  for (int p = n; p <= 4; ++p) ga.start[p] = start;
  ga.n = n;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return SD_ERR_UNSUPPORTED;
  cus &= ~7;
  if (cus <= 0) return SD_ERR_UNSUPPORTED;
  if (const int cb = cu_budget(cus, K); cb > 0 && cb < cus) cus = cb;
  SdProfScope prof(SD_K_GEMM_TN, flops, (hipStream_t)stream);
  SD_PROF_LABEL("gemm_pgroup_tn_kernel<%s>", accumulate ? "true" : "false");
  // (sharing the DMA issue with the compute waves, gemm_pstag_kernel's +3 %, was measured 2-4 % SLOWER here -- 73.5-74.6 vs
  // 76.3-78.3 us for a student layer's four weight gradients -- so the producers issue every piece: SHARE = false)
  const dim3 grid(start < cus ? start : cus);
#define SD_TN_GO(ACC, SH) hipLaunchKernelGGL((gemm_pgroup_tn_kernel<ACC, SH>), grid, dim3(768), 0, (hipStream_t)stream, ga, K, 4)
  if (accumulate) SD_TN_GO(true, false);
  else SD_TN_GO(false, false);
#undef SD_TN_GO
  SD_CHECK_LAUNCH();
  return 0;
}

// dX of the down projection with the SwiGLU backward in its epilogue: d(gate|up) [M,2I] from dy [M,h], W_down [h,I]
// and the forward's gate|up [M,2I]; d(act) is never stored.  Bit-identical to sd_gemm_bf16 (NN) + sd_swiglu_bwd.
extern "C" int sd_gemm_swiglu_bwd(const void* dy, const void* wdown, const void* gate_up, void* dgate_up, int M, int I,
                                  int H, void* stream) {
  if (M <= 0 || I <= 0 || H <= 0 || (I & 7) || (H & 7)) return SD_ERR_UNSUPPORTED;
  if (((uintptr_t)dy | (uintptr_t)wdown | (uintptr_t)gate_up | (uintptr_t)dgate_up) & 15) return SD_ERR_ALIGN;
  EpiArgs ea{};
  ea.out2 = (bf16*)dgate_up;
  ea.ld2 = 2L * I;
  ea.I = I;
  ea.g0 = (const bf16*)gate_up;
  // NN: C[M,I] = dy[M,H] . W[H,I]; C itself (d act) is not written: the epilogue needs a non-null C only for alignment checks
  return dispatch(dy, wdown, dgate_up, nullptr, nullptr, 1, M, I, H, H, I, I, 0, 0, 1, (hipStream_t)stream, 5, &ea);
}

// o-projection dX with delta = rowsum(dO * O) per (token, head) in its epilogue (the flash-attention backward's row
// constant; attn_delta_kernel otherwise): d_ao [M, Hq*128] = dy [M,H] . Wo [H, Hq*128]; o = the forward's attention output
// (row stride ldo); delta [B, Hq, T] fp32, token m = b*T + t.  One 128-column tile is exactly one head.
extern "C" int sd_gemm_odx_delta(const void* dy, const void* wo, void* d_ao, const void* o, int64_t ldo, float* delta, int M,
                                 int T, int Hq, int H, void* stream) {
  if (M <= 0 || T <= 0 || (M % T) || Hq <= 0 || H <= 0 || (H & 7) || (ldo & 7)) return SD_ERR_UNSUPPORTED;
  if (((uintptr_t)dy | (uintptr_t)wo | (uintptr_t)d_ao | (uintptr_t)o | (uintptr_t)delta) & 15) return SD_ERR_ALIGN;
  EpiArgs ea{};
  ea.out2 = (bf16*)delta;
  ea.g0 = (const bf16*)o;
  ea.ld2 = ldo;
  ea.T = T;
  ea.Hq = Hq;
  const int QD = Hq * 128;
  return dispatch(dy, wo, d_ao, nullptr, nullptr, 1, M, QD, H, H, QD, QD, 0, 0, 1, (hipStream_t)stream, 6, &ea);
}

// ---- folded RMSNorm (the frozen teacher's inference forward, sd_hip.h SD_SAVE_NONE_FOLDED): the norm's gain lives in the
// weight rows of the projection behind it, its row statistic travels as per-tile partial sums of squares [M, K / 128].
static bool ssq_shape_ok(int K) { return K > 0 && (K % 128) == 0 && K / 128 <= 16 && ((K / 128) % 4) == 0; }

// C [M,N] = A [M,K] . B [N,K]^T + R, and ssq_out [N/128, M] (tile-major) = per 128-column tile sums of squares of the stored bf16 C
extern "C" int sd_gemm_bf16_ssq(const void* A, const void* B, void* C, const void* R, float* ssq_out, int M, int N, int K,
                                int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, void* stream) {
  if (int e = check_args(A, B, C, R, M, N, K, lda, ldb, ldc, ldr, 0, 0)) return e;
  if (!R || !ssq_out || !ssq_shape_ok(N)) return SD_ERR_UNSUPPORTED;
  if ((uintptr_t)ssq_out & 15) return SD_ERR_ALIGN;
  EpiArgs ea{};
  ea.ssq_out = ssq_out;
  ea.ssq_n = N / 128;
  return dispatch(A, B, C, R, nullptr, 1, M, N, K, lda, ldb, ldc, ldr, 0, 0, (hipStream_t)stream, 0, &ea);
}

// gate|up projection with SwiGLU fused into the epilogue (HF:81-83): act [M,I] = silu(x Wg^T) * (x Wu^T);
// wgu = [gate rows | up rows] ([2I,K], torch layout); gu_out [M,2I] (gate | up, for the backward) may be NULL.
// ssq != NULL: x is the UN-normalised row, wgu carries the norm's gain, ssq [K/128, M] its partial sums of squares.
static int gemm_swiglu_impl(const void* x, const void* wgu, void* gu_out, void* act_out, const float* ssq, float eps, int M,
                            int I, int K, void* stream) {
  if (M <= 0 || I <= 0 || K <= 0 || (I % 64) || (K % BK)) return SD_ERR_UNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)wgu | (uintptr_t)gu_out | (uintptr_t)act_out | (uintptr_t)ssq) & 15) return SD_ERR_ALIGN;
  if (ssq && !ssq_shape_ok(K)) return SD_ERR_UNSUPPORTED;
  EpiArgs ea{};
  ea.out2 = (bf16*)act_out;
  ea.ld2 = I;
  ea.I = I;
  if (ssq) { ea.ssq_in = ssq; ea.ssq_n = K / 128; ea.inv_h = 1.f / (float)K; ea.eps_rs = eps; }
  return dispatch(x, wgu, gu_out, nullptr, nullptr, 1, M, 2 * I, K, K, K, 2 * I, 0, 0, 0, (hipStream_t)stream, 3, &ea);
}
extern "C" int sd_gemm_swiglu(const void* x, const void* wgu, void* gu_out, void* act_out, int M, int I, int K,
                              void* stream) {
  return gemm_swiglu_impl(x, wgu, gu_out, act_out, nullptr, 0.f, M, I, K, stream);
}
extern "C" int sd_gemm_swiglu_rs(const void* x, const void* wgu, void* gu_out, void* act_out, const float* ssq, float eps,
                                 int M, int I, int K, void* stream) {
  if (!ssq) return SD_ERR_SHAPE;
  return gemm_swiglu_impl(x, wgu, gu_out, act_out, ssq, eps, M, I, K, stream);
}

// q|k|v projection with the per-head q/k RMSNorm and rotate-half RoPE fused into the epilogue (HF:252-257, 121-170):
// qkv_out [M,(Hq+2Hkv)*128] raw projections (V for attention, q/k for the backward), qk_out [M,(Hq+Hkv)*128] rotated.
static int gemm_qkv_rope_impl(const void* x, const void* wqkv, void* qkv_out, void* qk_out, const void* q_gain,
                              const void* k_gain, const void* cos_tab, const void* sin_tab, const float* ssq, int M, int T,
                              int Hq, int Hkv, int K, float eps, void* stream);
extern "C" int sd_gemm_qkv_rope(const void* x, const void* wqkv, void* qkv_out, void* qk_out, const void* q_gain,
                                const void* k_gain, const void* cos_tab, const void* sin_tab, int M, int T, int Hq,
                                int Hkv, int K, float eps, void* stream) {
  return gemm_qkv_rope_impl(x, wqkv, qkv_out, qk_out, q_gain, k_gain, cos_tab, sin_tab, nullptr, M, T, Hq, Hkv, K, eps, stream);
}
// ssq [K/128, M]: x is the UN-normalised row and wqkv carries the input norm's gain (see sd_gemm_bf16_ssq)
extern "C" int sd_gemm_qkv_rope_rs(const void* x, const void* wqkv, void* qkv_out, void* qk_out, const void* q_gain,
                                   const void* k_gain, const void* cos_tab, const void* sin_tab, const float* ssq, int M,
                                   int T, int Hq, int Hkv, int K, float eps, void* stream) {
  if (!ssq) return SD_ERR_SHAPE;
  return gemm_qkv_rope_impl(x, wqkv, qkv_out, qk_out, q_gain, k_gain, cos_tab, sin_tab, ssq, M, T, Hq, Hkv, K, eps, stream);
}
static int gemm_qkv_rope_impl(const void* x, const void* wqkv, void* qkv_out, void* qk_out, const void* q_gain,
                              const void* k_gain, const void* cos_tab, const void* sin_tab, const float* ssq, int M, int T,
                              int Hq, int Hkv, int K, float eps, void* stream) {
  if (M <= 0 || T <= 0 || (M % T) || K <= 0 || (K % BK) || !qkv_out || !qk_out) return SD_ERR_UNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)wqkv | (uintptr_t)qkv_out | (uintptr_t)qk_out | (uintptr_t)ssq) & 15) return SD_ERR_ALIGN;
  if (ssq && !ssq_shape_ok(K)) return SD_ERR_UNSUPPORTED;
  EpiArgs ea{};
  if (ssq) { ea.ssq_in = ssq; ea.ssq_n = K / 128; ea.inv_h = 1.f / (float)K; ea.eps_rs = eps; }
  ea.out2 = (bf16*)qk_out;
  ea.ld2 = (long)(Hq + Hkv) * 128;
  ea.g0 = (const bf16*)q_gain;
  ea.g1 = (const bf16*)k_gain;
  ea.cos_t = (const bf16*)cos_tab;
  ea.sin_t = (const bf16*)sin_tab;
  ea.T = T;
  ea.Hq = Hq;
  ea.Hkv = Hkv;
  ea.eps = eps;
  const int N = (Hq + 2 * Hkv) * 128;
  return dispatch(x, wqkv, qkv_out, nullptr, nullptr, 1, M, N, K, K, K, N, 0, 0, 0, (hipStream_t)stream, 4, &ea);
}
