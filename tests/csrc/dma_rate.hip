// Measurement only (tests/bench_dma_rate.py): how fast can ONE CU fill LDS by LDS-DMA (buffer_load ... lds, 1 KiB per
// wave-instruction) when all 256 CUs stream the operand pattern of the 256x128x64 tile GEMM -- no MFMA, no LDS reads.
// A workgroup = NWAVES issuing waves; per step it stages STAGE_KB KiB: an "A panel" shared by the 8 workgroups of a tile
// row and a "B panel" shared by the 4 of a tile column (per XCD: 4 x 8 tiles), exactly the sharing of gemm_pstag_kernel,
// so that the L2 hit rate is the GEMM's (0.83).  Ring of NST stages, counted vmcnt, one barrier per step.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SD_LDS __attribute__((address_space(3)))

template <int NWAVES, int NST, int STAGE_KB>
__global__ __launch_bounds__(NWAVES * 64) void dma_rate_kernel(const char* __restrict__ A, const char* __restrict__ B,
                                                               long a_bytes, long b_bytes, int steps, int mode,
                                                               unsigned long long* out) {
  constexpr int PIECES = STAGE_KB;              // 1 KiB pieces per stage
  constexpr int PER = PIECES / NWAVES;          // per wave and stage
  static_assert(PIECES % NWAVES == 0, "pieces per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;  // j = 0..31 inside the XCD
  const int tm = j >> 3, tn = j & 7;
  // A panel of (xcd, tm): 2/3 of the stage; B panel of (xcd, tn): 1/3.  mode 1: private regions (no sharing)
  // Round 4: offsets are 32-bit and wrap with a MASK (a_bytes / b_bytes are powers of two).  The round-3 form took a
  // 64-bit `% a_bytes` per DMA piece, which hipcc expands to a software division (~130 instructions and three branches)
  // in front of EVERY buffer_load ... lds: its "63 GB/s per CU" was the rate of that issue stream, not of the load path.
  const unsigned a_mask = (unsigned)(a_bytes - 1), b_mask = (unsigned)(b_bytes - 1);
  const unsigned a_panel = (unsigned)(mode == 1 ? (int)blockIdx.x : (xcd * 4 + tm)) * (unsigned)steps * (unsigned)(STAGE_KB * 1024 * 2 / 3);
  const unsigned b_panel = (unsigned)(mode == 1 ? (int)blockIdx.x : (xcd * 8 + tn)) * (unsigned)steps * (unsigned)(STAGE_KB * 1024 / 3);
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(a_bytes > 0x7fffffffL ? 0x7fffffff : a_bytes), 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(b_bytes > 0x7fffffffL ? 0x7fffffff : b_bytes), 0x00020000);
  constexpr int A_PIECES = PIECES * 2 / 3;
  auto issue = [&](int step, char* stage) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int p = w * PER + i;
      if (p < A_PIECES) {
        const unsigned off = (a_panel + (unsigned)step * (unsigned)(A_PIECES * 1024) + (unsigned)(p * 1024)) & a_mask;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (SD_LDS void*)(stage + p * 1024), 16, lane * 16, (int)off, 0, 0);
      } else {
        const unsigned off = (b_panel + (unsigned)step * (unsigned)((PIECES - A_PIECES) * 1024) + (unsigned)((p - A_PIECES) * 1024)) & b_mask;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (SD_LDS void*)(stage + p * 1024), 16, lane * 16, (int)off, 0, 0);
      }
    }
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
  for (int d = 0; d < NST - 1; ++d) issue(d, smem + d * STAGE_KB * 1024);
  int nxt = NST - 1;
  for (int g = 0; g < steps; ++g) {
    issue(g + NST - 1, smem + nxt * STAGE_KB * 1024);
    nxt = (nxt == NST - 1) ? 0 : nxt + 1;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 1) * PER) : "memory");  // stage g has landed
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;  // 100 MHz ticks
}

template <int NWAVES, int NST, int STAGE_KB>
static int go(const void* A, const void* B, long a_bytes, long b_bytes, int steps, int mode, unsigned long long* out, void* stream) {
  const size_t lds = (size_t)NST * STAGE_KB * 1024;
  hipFuncSetAttribute((const void*)dma_rate_kernel<NWAVES, NST, STAGE_KB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((dma_rate_kernel<NWAVES, NST, STAGE_KB>), dim3(256), dim3(NWAVES * 64), lds, (hipStream_t)stream,
                     (const char*)A, (const char*)B, a_bytes, b_bytes, steps, mode, out);
  return (int)hipGetLastError();
}

extern "C" int dma_rate(const void* A, const void* B, long a_bytes, long b_bytes, int steps, int mode, int nwaves, int nst,
                        int stage_kb, unsigned long long* out, void* stream) {
#define CASE(W, S, K) if (nwaves == W && nst == S && stage_kb == K) return go<W, S, K>(A, B, a_bytes, b_bytes, steps, mode, out, stream)
  CASE(4, 3, 48); CASE(8, 3, 48); CASE(12, 3, 48); CASE(16, 3, 48);
  CASE(4, 6, 24); CASE(8, 6, 24); CASE(12, 6, 24);
  CASE(4, 2, 48); CASE(8, 2, 48);
  CASE(4, 12, 12); CASE(12, 12, 12);
  CASE(16, 9, 16);
#undef CASE
  return -1;
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: the SAME LDS-DMA panel stream PLUS a concurrent buffer_load_dwordx4 -> VGPR stream of the
// MFMA B fragments of an NT GEMM.  For v_mfma_f32_16x16x32_bf16 lane l of a B fragment holds weight row (l & 15), k =
// 8 (l >> 4) .. +7: 16 contiguous bytes of a [N][K] weight -- it needs no LDS at all.  Roles as in gemm_pstag_kernel:
// NPROD producer waves stage the A panel (A_KB KiB per 64-deep K-step) by LDS-DMA into a ring, NCOMP "compute" waves
// each load the fragments of BROWS weight rows x 64 k (BROWS / 16 x 2 loads per lane and step) one K-step ahead into a
// second register set, fold them into a checksum (no MFMA, no LDS reads) and meet the producers at one barrier per
// step.  Waves w and w + wn_groups load the SAME rows (the duplication of an (m x n) wave grid: 4 x 2 waves on a 256 x 128
// tile = wn_groups 2, each B row fetched by 4 waves).  B panel of (xcd, tn) shared by the 4 workgroups of a tile column,
// A panel of (xcd, tm) by 8, as above.  out[blockIdx.x] = 100 MHz ticks; sink takes the checksum (never all-equal).
template <int NPROD, int NCOMP, int NST, int A_KB, int BROWS>
__global__ __launch_bounds__((NPROD + NCOMP) * 64) void mix_rate_kernel(const char* __restrict__ A, const char* __restrict__ B,
                                                                        long a_bytes, long b_bytes, int steps, int wn_groups,
                                                                        int n_cols, unsigned long long* out, unsigned* sink) {
  constexpr int NJ = BROWS / 16, BL = NJ * 2;   // loads per lane and K-step
  constexpr int PER = (A_KB > 0 && NPROD > 0) ? A_KB / NPROD : 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int tm = j >> 3, tn = j & 7;
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(a_bytes > 0x7fffffffL ? 0x7fffffff : a_bytes), 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(b_bytes > 0x7fffffffL ? 0x7fffffff : b_bytes), 0x00020000);
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (w < NPROD) {
    if constexpr (PER > 0) {
      const unsigned a_mask = (unsigned)(a_bytes - 1);
      const unsigned a_panel = (unsigned)(xcd * 4 + tm) * (unsigned)steps * (unsigned)(A_KB * 1024);
      auto issue = [&](int step, char* stage) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
          const int p = w * PER + i;
          const unsigned off = (a_panel + (unsigned)step * (unsigned)(A_KB * 1024) + (unsigned)(p * 1024)) & a_mask;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (SD_LDS void*)(stage + p * 1024), 16, lane * 16, (int)off, 0, 0);
        }
      };
#pragma unroll
      for (int d = 0; d < NST - 1; ++d) issue(d, smem + d * A_KB * 1024);
      int nxt = NST - 1;
      for (int g = 0; g < steps; ++g) {
        issue(g + NST - 1, smem + nxt * A_KB * 1024);
        nxt = (nxt == NST - 1) ? 0 : nxt + 1;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 1) * PER) : "memory");
        __builtin_amdgcn_s_barrier();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      for (int g = 0; g < steps; ++g) __builtin_amdgcn_s_barrier();
    }
  } else {
    // weight rows of this wave: [N][K] bf16 with K = 64 * steps (row stride ldb bytes); the tile column's n0, then the
    // wave's row group
    const int cw = w - NPROD;
    const long ldb = (long)steps * 128;  // bytes per weight row
    const long row0 = ((long)(xcd * 8 + tn) * n_cols + (long)(cw % wn_groups) * BROWS) % (b_bytes / ldb - BROWS);
    int voff[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) voff[jj] = (int)((row0 + jj * 16 + (lane & 15)) * ldb + (lane >> 4) * 16);
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    u4 cur[BL], nxt[BL];
    auto load = [&](u4 (&r)[BL], int step) __attribute__((always_inline)) {
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          r[jj * 2 + kk] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rb, voff[jj], step * 128 + kk * 64, 0));
    };
    unsigned acc = 0;
    load(cur, 0);
    for (int g = 0; g < steps; g += 2) {
      load(nxt, g + 1 < steps ? g + 1 : g);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BL) : "memory");
#pragma unroll
      for (int i = 0; i < BL; ++i) acc ^= cur[i][0] ^ cur[i][1] ^ cur[i][2] ^ cur[i][3];
      __builtin_amdgcn_s_barrier();
      load(cur, g + 2 < steps ? g + 2 : g);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BL) : "memory");
#pragma unroll
      for (int i = 0; i < BL; ++i) acc ^= nxt[i][0] ^ nxt[i][1] ^ nxt[i][2] ^ nxt[i][3];
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) sink[blockIdx.x * 64 + lane] = acc;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int NPROD, int NCOMP, int NST, int A_KB, int BROWS>
static int go_mix(const void* A, const void* B, long a_bytes, long b_bytes, int steps, int wn_groups, int n_cols,
                  unsigned long long* out, unsigned* sink, void* stream) {
  const size_t lds = (size_t)NST * (A_KB > 0 ? A_KB : 1) * 1024;
  hipFuncSetAttribute((const void*)mix_rate_kernel<NPROD, NCOMP, NST, A_KB, BROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((mix_rate_kernel<NPROD, NCOMP, NST, A_KB, BROWS>), dim3(256), dim3((NPROD + NCOMP) * 64), lds,
                     (hipStream_t)stream, (const char*)A, (const char*)B, a_bytes, b_bytes, steps, wn_groups, n_cols, out, sink);
  return (int)hipGetLastError();
}

// steps must be even.  b_bytes / (128 * steps) = number of weight rows available.
extern "C" int mix_rate(const void* A, const void* B, long a_bytes, long b_bytes, int steps, int nprod, int ncomp, int nst,
                        int a_kb, int brows, int wn_groups, int n_cols, unsigned long long* out, unsigned* sink, void* stream) {
#define CASE(P, Cw, S, K, R) if (nprod == P && ncomp == Cw && nst == S && a_kb == K && brows == R) \
    return go_mix<P, Cw, S, K, R>(A, B, a_bytes, b_bytes, steps, wn_groups, n_cols, out, sink, stream)
  CASE(4, 8, 3, 32, 64);   // 256 x 128 tile: A by DMA (4 producers), 8 waves x 64 B rows (4 x 2 wave grid: wn_groups 2)
  CASE(4, 8, 3, 32, 32);   // 2 x 4 wave grid: 128 x 32 per wave (wn_groups 4)
  CASE(4, 8, 3, 32, 16);   // 1 x 8 wave grid: 256 x 16 per wave (wn_groups 8): no duplication
  CASE(0, 8, 3, 0, 64);    // VGPR stream alone
  CASE(0, 8, 3, 0, 32);
  CASE(0, 8, 3, 0, 16);
  CASE(4, 8, 3, 48, 16);   // today's full DMA stream + a little VGPR traffic
  CASE(4, 4, 3, 32, 128);  // 256 x 256 tile, 4 compute waves of 128 x 128 (wn_groups 2): A by DMA, B direct
  CASE(4, 4, 3, 32, 64);   // 256 x 128 tile, 4 compute waves of 128 x 64
  CASE(8, 8, 3, 32, 64);
  CASE(4, 8, 5, 16, 64);   // 128-row A panel
#undef CASE
  return -1;
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4, third arm: the SAME panel stream staged the classic way -- buffer_load_dwordx4 to VGPRs (fully coalesced: a
// wave-instruction reads 1 KiB contiguous, 8 lanes per 128-byte line) and ds_write_b128 into the LDS ring (write = 1) or
// dropped into a checksum (write = 0: the pure global -> VGPR rate of this access pattern).  DEPTH steps of loads in
// flight per wave (registers: DEPTH x PER x 4), one barrier per step.  Same sharing as dma_rate_kernel mode 0.
template <int NWAVES, int DEPTH, int STAGE_KB>
__global__ __launch_bounds__(NWAVES * 64) void stage_rate_kernel(const char* __restrict__ A, const char* __restrict__ B,
                                                                 long a_bytes, long b_bytes, int steps, int write,
                                                                 unsigned long long* out, unsigned* sink) {
  constexpr int PIECES = STAGE_KB, PER = PIECES / NWAVES, A_PIECES = PIECES * 2 / 3;
  static_assert(PIECES % NWAVES == 0 && (DEPTH == 1 || DEPTH == 2), "shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int tm = j >> 3, tn = j & 7;
  const unsigned a_mask = (unsigned)(a_bytes - 1), b_mask = (unsigned)(b_bytes - 1);
  const unsigned a_panel = (unsigned)(xcd * 4 + tm) * (unsigned)steps * (unsigned)(STAGE_KB * 1024 * 2 / 3);
  const unsigned b_panel = (unsigned)(xcd * 8 + tn) * (unsigned)steps * (unsigned)(STAGE_KB * 1024 / 3);
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(a_bytes > 0x7fffffffL ? 0x7fffffff : a_bytes), 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(b_bytes > 0x7fffffffL ? 0x7fffffff : b_bytes), 0x00020000);
  auto load = [&](u4 (&r)[PER], int step) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int p = w * PER + i;
      if (p < A_PIECES) {
        const unsigned off = (a_panel + (unsigned)step * (unsigned)(A_PIECES * 1024) + (unsigned)(p * 1024)) & a_mask;
        r[i] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(ra, lane * 16, (int)off, 0));
      } else {
        const unsigned off = (b_panel + (unsigned)step * (unsigned)((PIECES - A_PIECES) * 1024) + (unsigned)((p - A_PIECES) * 1024)) & b_mask;
        r[i] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rb, lane * 16, (int)off, 0));
      }
    }
  };
  unsigned acc = 0;
  auto consume = [&](u4 (&r)[PER], char* stage) __attribute__((always_inline)) {
    if (write) {
#pragma unroll
      for (int i = 0; i < PER; ++i) *(u4*)(stage + (w * PER + i) * 1024 + lane * 16) = r[i];
    } else {
#pragma unroll
      for (int i = 0; i < PER; ++i) acc ^= r[i][0] ^ r[i][1] ^ r[i][2] ^ r[i][3];
    }
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  u4 r0[PER], r1[PER];
  load(r0, 0);
  if constexpr (DEPTH == 2) load(r1, 1);
  for (int g = 0; g < steps; g += 2) {
    if constexpr (DEPTH == 2) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
      consume(r0, smem);
      load(r0, g + 2);
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
      consume(r1, smem + STAGE_KB * 1024);
      load(r1, g + 3);
      __builtin_amdgcn_s_barrier();
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      consume(r0, smem);
      load(r0, g + 1);
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      consume(r0, smem + STAGE_KB * 1024);
      load(r0, g + 2);
      __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (acc == 0x12345678u) sink[blockIdx.x * 64 + lane] = acc + r0[0][0] + (DEPTH == 2 ? r1[0][0] : 0u);
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int NWAVES, int DEPTH, int STAGE_KB>
static int go_stage(const void* A, const void* B, long a_bytes, long b_bytes, int steps, int write, unsigned long long* out,
                    unsigned* sink, void* stream) {
  const size_t lds = (size_t)2 * STAGE_KB * 1024;
  hipFuncSetAttribute((const void*)stage_rate_kernel<NWAVES, DEPTH, STAGE_KB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((stage_rate_kernel<NWAVES, DEPTH, STAGE_KB>), dim3(256), dim3(NWAVES * 64), lds, (hipStream_t)stream,
                     (const char*)A, (const char*)B, a_bytes, b_bytes, steps, write, out, sink);
  return (int)hipGetLastError();
}

extern "C" int stage_rate(const void* A, const void* B, long a_bytes, long b_bytes, int steps, int write, int nwaves, int depth,
                          int stage_kb, unsigned long long* out, unsigned* sink, void* stream) {
#define CASE(W, D, K) if (nwaves == W && depth == D && stage_kb == K) return go_stage<W, D, K>(A, B, a_bytes, b_bytes, steps, write, out, sink, stream)
  CASE(4, 1, 48); CASE(4, 2, 48); CASE(8, 1, 48); CASE(8, 2, 48); CASE(12, 2, 48); CASE(16, 2, 48);
  CASE(4, 2, 24); CASE(8, 2, 24);
  CASE(4, 1, 64); CASE(4, 2, 64); CASE(8, 2, 64);
#undef CASE
  return -1;
}
