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
  const long a_panel = (mode == 1 ? (long)blockIdx.x : (long)(xcd * 4 + tm)) * (long)steps * (STAGE_KB * 1024L * 2 / 3);
  const long b_panel = (mode == 1 ? (long)blockIdx.x : (long)(xcd * 8 + tn)) * (long)steps * (STAGE_KB * 1024L / 3);
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(a_bytes > 0x7fffffffL ? 0x7fffffff : a_bytes), 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(b_bytes > 0x7fffffffL ? 0x7fffffff : b_bytes), 0x00020000);
  constexpr int A_PIECES = PIECES * 2 / 3;
  auto issue = [&](int step, char* stage) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int p = w * PER + i;
      if (p < A_PIECES) {
        const long off = (a_panel + (long)step * (A_PIECES * 1024L) + p * 1024L) % a_bytes;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (SD_LDS void*)(stage + p * 1024), 16, lane * 16, (int)off, 0, 0);
      } else {
        const long off = (b_panel + (long)step * ((PIECES - A_PIECES) * 1024L) + (p - A_PIECES) * 1024L) % b_bytes;
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
