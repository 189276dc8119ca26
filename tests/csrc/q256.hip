// EXPERIMENT (tests/bench_q256.py; not part of libsd_hip.so): a 256 x 256 x 64 tile held by FOUR waves, one per SIMD, each
// owning 128 x 128 of C (256 accumulator registers, in AGPRs: this file is built WITHOUT -amdgpu-mfma-vgpr-form), operands
// staged the classic way -- fully coalesced buffer_load_dwordx4 into VGPRs one K-step ahead, ds_write_b128 into a
// two-stage LDS ring -- instead of LDS-DMA: the structure of the vendor's Custom_Cijk...MT256x256x64 kernel, which the
// round-4 yardstick measured 11-20 % faster than gemm_p256_kernel (8 waves of 64 x 128, self-issued LDS-DMA) on the
// config-5 shapes (profiles/r04_gemm_yardstick.json).  NT only: C [M,N] = A [M,K] . B [N,K]^T, K % 64 == 0.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../speech_distill_amd/csrc/sd_common.cuh"

extern "C" __device__ __attribute__((aligned(256))) unsigned char sd_zero_page[1024] = {0};

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int A_BYTES = BM * BK * 2, STAGE = A_BYTES + BN * BK * 2;  // 32 + 32 KiB

SD_DEV void tile_coords(int tile, int tiles_m, int tiles_n, int group_m, int& tm, int& tn) {
  const int per_group = group_m * tiles_n;
  const int g = tile / per_group, r = tile - g * per_group, g0 = g * group_m;
  const int gh = min(group_m, tiles_m - g0);
  tn = r / gh;
  tm = g0 + (r - tn * gh);
}

// v_mfma_f32_16x16x32_bf16 with the accumulator PINNED to AGPRs ("+a"): with the builtin hipcc spread the 256 accumulator
// registers of a wave over both files and moved them around every block (324 v_accvgpr_* per K-step, 4 spills).
SD_DEV void mfma_a(const bf16x8& b, const bf16x8& a, f32x4& c) {
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(b), "v"(a));
}

// 16 rows x 32 k fragment of a [rows][64 k] image (128-byte rows, 16-byte chunks XOR-swizzled by row & 7)
template <int MODE = 0>
SD_DEV bf16x8 frag(const char* img, int row16, int kk, int lane) {
  if constexpr (MODE == 2) { bf16x8 z; asm volatile("" : "=v"(z)); return z; }
  const int r = row16 + (lane & 15), c = kk * 4 + (lane >> 4);
  return *(const bf16x8*)(img + r * 128 + ((c ^ (r & 7)) << 4));
}

template <int PERSIST, int MODE>  // MODE (diagnosis): 0 full; 1 no staging traffic in the loop; 2 no fragment reads either; 3 no barrier; 4 ds_write but no global loads; 5 global loads but no ds_write
__global__ __launch_bounds__(256) void q256_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* __restrict__ C,
                                                   int M, int N, int K, long lda, long ldb, long ldc, int tiles_m, int tiles_n,
                                                   int group_m) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];  // 128 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int ntiles = tiles_m * tiles_n;
  const int nk = K / BK;
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(((long)(M - 1) * lda + K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(((long)(N - 1) * ldb + K) * 2), 0x00020000);
#endif
  // staging: load i of a thread covers image row i*32 + (tid >> 3), chunk tid & 7 (8 lanes = one 128-byte row)
  const int srow = tid >> 3, sc = tid & 7;
  const unsigned lds_off = (unsigned)(srow * 128 + ((sc ^ (srow & 7)) << 4));
  char* ep = smem + w * 2048;  // epilogue patch of this wave (the ring is free by then)

  for (int tile = (int)blockIdx.x; tile < ntiles; tile += PERSIST ? (int)gridDim.x : ntiles) {
    int tm, tn;
    tile_coords(xcd_remap(tile, ntiles), tiles_m, tiles_n, group_m, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int va = (int)(((long)(m0 + srow) * lda + sc * 8) * 2), vb = (int)(((long)(n0 + srow) * ldb + sc * 8) * 2);
    // ONE staging register set used as a rolling pipeline: in step t register i is written to LDS (K-step t+1) at its slot
    // and at once re-loaded from global memory (K-step t+2), so every load has a whole K-step to land and the 16
    // ds_write_b128 / buffer_load pairs of a step are spread over 96 MFMAs.  (Two full sets spilled: 381 registers.)
    u4 sa[1][8], sb[1][8];
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#define FENCE() __builtin_amdgcn_sched_barrier(0)
    auto gload1 = [&](int set, int i, int k0) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
      if (i < 8) sa[set][i] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(ra, va, (int)(((long)i * 32 * lda + k0) * 2), 0));
      else sb[set][i - 8] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rb, vb, (int)(((long)(i - 8) * 32 * ldb + k0) * 2), 0));
#endif
    };
    auto lstore1 = [&](int set, int i, char* stage) __attribute__((always_inline)) {
      if (i < 8) *(u4*)(stage + lds_off + i * 4096) = sa[set][i];
      else *(u4*)(stage + A_BYTES + lds_off + (i - 8) * 4096) = sb[set][i - 8];
    };
    // Software pipeline of a K-step t (stage t&1; fragments double-buffered in registers, every memory instruction placed
    // between two MFMAs by hand -- the asm MFMAs are opaque to sched_group_barrier, so order is pinned with sched_barrier):
    //   block 0  acc[0..3] += A(kk0, rows 0-63)   . B(kk0)   | reads A(kk0, rows 64-127), B(kk1)[0..3]
    //   block 1  acc[4..7] += A(kk0, rows 64-127) . B(kk0)   | reads A(kk1, rows 0-63), B(kk1)[4..7]
    //   block 2  acc[0..3] += A(kk1, rows 0-63)   . B(kk1)   | reads A(kk1, rows 64-127)
    //   blocks 0-2: staging register i: ds_write_b128 (K-step t+1 -> stage (t+1)&1), then buffer_load (K-step t+2); one pair
    //               per 5-6 MFMAs
    //   lgkmcnt(0); barrier   (stage (t+1)&1 complete; every read of stage t&1 has been issued and returned)
    //   block 3  acc[4..7] += A(kk1, rows 64-127) . B(kk1)   | reads A, B (kk0, rows 0-63) of K-step t+1
    // Steps past the end load / store harmless data (clamped by the buffer range check; nobody reads that stage).
    __syncthreads();  // the previous tile's epilogue patches are done with the ring
#pragma unroll
    for (int i = 0; i < 16; ++i) gload1(0, i, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) { lstore1(0, i, smem); gload1(0, i, BK); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bf16x8 a0[4], a1[4], b0[8], b1[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) a0[i] = frag(smem, wm * 128 + i * 16, 0, lane);
#pragma unroll
    for (int j = 0; j < 8; ++j) b0[j] = frag(smem + A_BYTES, wn * 128 + j * 16, 0, lane);
    // set holding K-step t+1 = (t+1)&1 ; set to fill with K-step t+2 = t&1 (its data, K-step t, went to LDS during step t-1)
    auto step = [&](int par, int t) __attribute__((always_inline)) {
      const char* ai = smem + par * STAGE;
      const char* bi = ai + A_BYTES;
      char* nx = smem + (par ^ 1) * STAGE;
      FENCE();
#pragma unroll
      for (int n = 0; n < 32; ++n) {  // block 0
        mfma_a(b0[n & 7], a0[n >> 3], acc[n >> 3][n & 7]);
        if ((n & 3) == 0) a1[n >> 2 & 3] = frag<MODE>(ai, wm * 128 + 64 + (n >> 2 & 3) * 16, 0, lane);
        if (n < 16 && (n & 3) == 2) b1[n >> 2] = frag<MODE>(bi, wn * 128 + (n >> 2) * 16, 1, lane);
        if (n % 6 == 5) { if (MODE != 1 && MODE != 2 && MODE != 5) lstore1(0, n / 6, nx); if (MODE != 1 && MODE != 2 && MODE != 4) gload1(0, n / 6, (t + 2) * BK); }  // registers 0..4
        FENCE();
      }
#pragma unroll
      for (int n = 0; n < 32; ++n) {  // block 1
        mfma_a(b0[n & 7], a1[n >> 3], acc[4 + (n >> 3)][n & 7]);
        if (n < 16 && (n & 3) == 0) a0[n >> 2] = frag<MODE>(ai, wm * 128 + (n >> 2) * 16, 1, lane);
        if (n < 16 && (n & 3) == 2) b1[4 + (n >> 2)] = frag<MODE>(bi, wn * 128 + (4 + (n >> 2)) * 16, 1, lane);
        if (n % 6 == 5) { if (MODE != 1 && MODE != 2 && MODE != 5) lstore1(0, 5 + n / 6, nx); if (MODE != 1 && MODE != 2 && MODE != 4) gload1(0, 5 + n / 6, (t + 2) * BK); }  // registers 5..9
        FENCE();
      }
#pragma unroll
      for (int n = 0; n < 32; ++n) {  // block 2
        mfma_a(b1[n & 7], a0[n >> 3], acc[n >> 3][n & 7]);
        if (n < 16 && (n & 3) == 0) a1[n >> 2] = frag<MODE>(ai, wm * 128 + 64 + (n >> 2) * 16, 1, lane);
        if (n % 5 == 4 && n / 5 < 6) { if (MODE != 1 && MODE != 2 && MODE != 5) lstore1(0, 10 + n / 5, nx); if (MODE != 1 && MODE != 2 && MODE != 4) gload1(0, 10 + n / 5, (t + 2) * BK); }  // registers 10..15
        FENCE();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (MODE != 3) __builtin_amdgcn_s_barrier();
      FENCE();
#pragma unroll
      for (int n = 0; n < 32; ++n) {  // block 3
        if (MODE == 6 && n == 26) {
          // L2 prefetch of K-step t+PF: one dword of every 128-byte row of the A and B stage (64 rows per wave each);
          // the value is never used (asm: no wait is ever inserted for it)
          unsigned dummy;
          const int ra_ = min(m0 + w * 64 + lane, M - 1), rb_ = min(n0 + w * 64 + lane, N - 1);
          const int pk = min((t + 4) * BK, K - BK);
          const bf16* pa = A + (long)ra_ * lda + pk;
          const bf16* pb = B + (long)rb_ * ldb + pk;
          asm volatile("global_load_dword %0, %1, off" : "=v"(dummy) : "v"(pa));
          asm volatile("global_load_dword %0, %1, off" : "=v"(dummy) : "v"(pb));
        }
        mfma_a(b1[n & 7], a1[n >> 3], acc[4 + (n >> 3)][n & 7]);
        if (n < 8 && (n & 1) == 0) a0[n >> 1] = frag<MODE>(nx, wm * 128 + (n >> 1) * 16, 0, lane);
        if (n >= 8 && n < 24 && (n & 1) == 0) b0[(n - 8) >> 1] = frag<MODE>(nx + A_BYTES, wn * 128 + ((n - 8) >> 1) * 16, 0, lane);
        FENCE();
      }
    };
    for (int t = 0; t < nk; ++t) step(t & 1, t);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the ring is free for the epilogue patches
#undef FENCE
    // the asm MFMAs are invisible to the compiler's hazard recognizer: cover the MFMA -> accumulator-read distance by hand
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    // epilogue: 16 rows x 64 columns at a time through the wave's LDS patch, whole 128-byte lines out
    const int r = lane & 15, q4 = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16)acc[i][jh * 4 + j][e];
          *(bf16x4*)(ep + r * 128 + (((2 * j + (q4 >> 1)) ^ (r & 7)) << 4) + (q4 & 1) * 8) = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int rr = hh * 8 + (lane >> 3), cc = lane & 7;
          const bf16x8 v = *(const bf16x8*)(ep + rr * 128 + ((cc ^ (rr & 7)) << 4));
          const int gm = m0 + wm * 128 + i * 16 + rr, gn = n0 + wn * 128 + jh * 64 + cc * 8;
          if (gm < M && gn < N) *(bf16x8*)(C + (long)gm * ldc + gn) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
  }
}

}  // namespace

extern "C" int q256_gemm(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc, int persist,
                         void* stream) {
  const int mode = persist >> 4;
  persist &= 15;
  if (K % BK || (N & 7)) return -1;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int gm = tiles_m < 8 ? tiles_m : 8;
  if (persist) {
    const int grid = tiles_m * tiles_n < 256 ? tiles_m * tiles_n : 256;
#define GO(MD) hipLaunchKernelGGL((q256_kernel<1, MD>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)A, (const bf16*)B, (bf16*)C, M, N, K, lda, ldb, ldc, tiles_m, tiles_n, gm)
    if (mode == 1) GO(1); else if (mode == 2) GO(2); else if (mode == 3) GO(3); else if (mode == 4) GO(4); else if (mode == 5) GO(5); else if (mode == 6) GO(6); else GO(0);
#undef GO
  } else {
    hipLaunchKernelGGL((q256_kernel<0, 0>), dim3(tiles_m * tiles_n), dim3(256), 0, (hipStream_t)stream, (const bf16*)A, (const bf16*)B,
                       (bf16*)C, M, N, K, lda, ldb, ldc, tiles_m, tiles_n, gm);
  }
  return (int)hipGetLastError();
}
