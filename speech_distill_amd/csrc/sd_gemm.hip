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
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16.
// Operand tiles go HBM -> LDS by 16-byte LDS-DMA (global_load_lds_dwordx4), double buffered; the
// LDS image is lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE address
// and undone on the fragment read.  K-contiguous operands are read with ds_read_b128, operands
// whose contraction index is the slow one with ds_read_b64_tr_b16 (hardware transpose).  The
// accumulator is produced transposed (mfma(Bfrag, Afrag)) so a lane owns 4 consecutive n; it is
// staged through LDS as fp32 and written out in full 16-byte row pieces with the epilogue fused.
#include "sd_common.cuh"
#include "../../include/sd_hip.h"

extern "C" __device__ __attribute__((aligned(256))) unsigned char sd_zero_page[1024] = {0};

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per stage

SD_DEV int swz_t(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// Stage one 128 x 64 operand tile into LDS.  TX=false: operand stored [rows][K] (K contiguous).
// TX=true: operand stored [K][rows] (rows contiguous).
template <bool TX>
SD_DEV void stage_tile(const bf16* __restrict__ g, long ld, int row0, int k0, int row_lim, int K,
                       char* lds_tile, int w, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = (w * 4 + i) * 64 + lane;
    const bf16* src;
    if constexpr (!TX) {
      const int r = p >> 3, s = p & 7, c = s ^ (r & 7);
      const int gr = row0 + r, gk = k0 + c * 8;
      src = g + (long)gr * ld + gk;
      if (gr >= row_lim || gk >= K) src = (const bf16*)(sd_zero_page + lane * 16);
    } else {
      const int k = p >> 4, u = p & 15;
      const int ch = (u >> 1) ^ swz_t(k);
      const int gc = row0 + ch * 16 + (u & 1) * 8, gk = k0 + k;
      src = g + (long)gk * ld + gc;
      if (gk >= K || gc >= row_lim) src = (const bf16*)(sd_zero_page + lane * 16);
    }
    glds16(src, lds_tile + (w * 4 + i) * 1024);
  }
}

// Fragment of 16 rows x 32 k for v_mfma_f32_16x16x32_bf16: lane l holds row (l&15), k = 8(l>>4)+j.
template <bool TX>
SD_DEV bf16x8 load_frag(const char* lds_tile, int row16_base, int kk, int lane) {
  if constexpr (!TX) {
    const int r = row16_base + (lane & 15);
    const int c = kk * 4 + (lane >> 4);
    return *(const bf16x8*)(lds_tile + ((r * 8 + (c ^ (r & 7))) << 4));
  } else {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    const int ch = row16_base >> 4;
    const int k0 = kk * 32 + 8 * g + q;
    const int k1 = k0 + 4;
    bf16x4 lo = lds_tr16(lds_tile + k0 * 256 + ((ch ^ swz_t(k0)) << 5) + 8 * pp);
    bf16x4 hi = lds_tr16(lds_tile + k1 * 256 + ((ch ^ swz_t(k1)) << 5) + 8 * pp);
    return cat8(lo, hi);
  }
}

template <bool TA, bool TB, bool HAS_R>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(
    const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* C, const bf16* R,
    int M, int N, int K, long lda, long ldb, long ldc, long ldr, int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // 2 stages x (A,B) = 64 KiB
  const int lane = lane_id();
  const int w = wave_id_uniform();
  const int wm = w >> 1, wn = w & 1;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (K + BK - 1) / BK;
  stage_tile<TA>(A, lda, m0, 0, M, K, smem, w, lane);
  stage_tile<TB>(B, ldb, n0, 0, N, K, smem + TILE_BYTES, w, lane);
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    char* cur = smem + (t & 1) * 2 * TILE_BYTES;
    if (t + 1 < nk) {
      char* nxt = smem + ((t + 1) & 1) * 2 * TILE_BYTES;
      stage_tile<TA>(A, lda, m0, (t + 1) * BK, M, K, nxt, w, lane);
      stage_tile<TB>(B, ldb, n0, (t + 1) * BK, N, K, nxt + TILE_BYTES, w, lane);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = load_frag<TA>(cur, wm * 64 + i * 16, kk, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = load_frag<TB>(cur + TILE_BYTES, wn * 64 + j * 16, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(bfr[j], af[i], acc[i][j]);  // D[n][m]: lane owns 4 consecutive n
    }
    __syncthreads();  // waits the LDS-DMA of tile t+1 (vmcnt(0)) and fences the reads of tile t
  }

  // Epilogue: fp32 tile -> LDS (XOR-swizzled 16-byte chunks), then coalesced bf16 rows out.
  float* cs = (float*)smem;  // [128][128] fp32 = 64 KiB
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
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int q = it * 256 + threadIdx.x;
    const int m = q >> 4, oc = q & 15;
    const int gm = m0 + m, gn = n0 + oc * 8;
    if (gm < M && gn < N) {
      f32x4 lo = *(const f32x4*)(cs + m * 128 + (((2 * oc) ^ (m & 15)) << 2));
      f32x4 hi = *(const f32x4*)(cs + m * 128 + (((2 * oc + 1) ^ (m & 15)) << 2));
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if constexpr (HAS_R) {
        bf16x8 r = *(const bf16x8*)(R + (long)gm * ldr + gn);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
      *(bf16x8*)(C + (long)gm * ldc + gn) = o;
    }
  }
}

template <bool TA, bool TB>
int launch(const void* A, const void* B, void* C, const void* R, int M, int N, int K, long lda, long ldb, long ldc,
           long ldr, hipStream_t st) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  dim3 grid(tiles_m * tiles_n), block(256);
  if (R)
    hipLaunchKernelGGL((gemm_bf16_kernel<TA, TB, true>), grid, block, 0, st, (const bf16*)A, (const bf16*)B, (bf16*)C,
                       (const bf16*)R, M, N, K, lda, ldb, ldc, ldr, tiles_m, tiles_n);
  else
    hipLaunchKernelGGL((gemm_bf16_kernel<TA, TB, false>), grid, block, 0, st, (const bf16*)A, (const bf16*)B, (bf16*)C,
                       (const bf16*)nullptr, M, N, K, lda, ldb, ldc, ldr, tiles_m, tiles_n);
  SD_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int sd_gemm_bf16(const void* A, const void* B, void* C, const void* R, int M, int N, int K, int64_t lda,
                            int64_t ldb, int64_t ldc, int64_t ldr, int trans_a, int trans_b, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return SD_ERR_SHAPE;
  if ((lda | ldb | ldc | (R ? ldr : 0)) & 7) return SD_ERR_ALIGN;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)R) & 15) return SD_ERR_ALIGN;
  if (N & 7) return SD_ERR_ALIGN;
  if ((!trans_a || !trans_b) && (K & 7)) return SD_ERR_ALIGN;
  if (trans_a && (M & 7)) return SD_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (!trans_a && !trans_b) return launch<false, false>(A, B, C, R, M, N, K, lda, ldb, ldc, ldr, st);
  if (!trans_a && trans_b) return launch<false, true>(A, B, C, R, M, N, K, lda, ldb, ldc, ldr, st);
  if (trans_a && trans_b) return launch<true, true>(A, B, C, R, M, N, K, lda, ldb, ldc, ldr, st);
  return launch<true, false>(A, B, C, R, M, N, K, lda, ldb, ldc, ldr, st);
}
