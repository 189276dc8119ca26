// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels of the distillation hot path.
// Wave = 64 lanes everywhere.  No CUDA-compat paths: this file only builds for gfx950.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define SD_LDS __attribute__((address_space(3)))
#define SD_GLB __attribute__((address_space(1)))
#define SD_DEV __device__ __forceinline__

constexpr int SD_WAVE = 64;

// A 1 KiB page of zeros in device memory: out-of-range lanes of an LDS-DMA tile load read from
// here, so tile edges (rows >= M, k >= K) are zero-filled without a branch around the load.
extern "C" __device__ __attribute__((aligned(256))) unsigned char sd_zero_page[1024];

SD_DEV int lane_id() { return threadIdx.x & 63; }
SD_DEV int wave_id_uniform() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

SD_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Cross-lane moves inside a 16-lane DPP row as VALU operands (v_*_dpp), instead of __shfl_xor, which hipcc lowers to
// ds_bpermute_b32 -- an LDS round trip per call (128 of them sat in the q|k|v epilogue).  Bit-identical replacements.
template <int CTRL>
SD_DEV float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over the 16 lanes of a row, result in every lane: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror,
// row_mirror = the xor 1, 2, 4, 8 butterfly (after two steps all lanes of a quad agree, after three all of a half-row)
SD_DEV float row16_sum(float v) {
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  v += dpp_move<0x141>(v);
  v += dpp_move<0x140>(v);
  return v;
}
// value of lane (i ^ 8) of the same row: row_ror:8
SD_DEV float row16_xor8(float v) { return dpp_move<0x128>(v); }

SD_DEV float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
SD_DEV int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide reductions through a small LDS scratch (>= 32 floats).  All threads get the result.
template <int NT>
SD_DEV float block_sum(float v, float* scratch) {
  constexpr int NW = NT / 64;
  v = wave_sum(v);
  if constexpr (NW == 1) return v;
  __syncthreads();
  if (lane_id() == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) r += scratch[i];
  return r;
}
template <int NT>
SD_DEV float block_max(float v, float* scratch) {
  constexpr int NW = NT / 64;
  v = wave_max(v);
  if constexpr (NW == 1) return v;
  __syncthreads();
  if (lane_id() == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = scratch[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) r = fmaxf(r, scratch[i]);
  return r;
}

// 16-byte global -> LDS DMA (global_load_lds_dwordx4).  The LDS destination is
// wave-uniform base + lane*16; the global source is per lane.
SD_DEV void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const SD_GLB void*)gsrc, (SD_LDS void*)lds_wave_base, 16, 0, 0);
}

// ds_read_b64_tr_b16: per 16-lane group, reads a 4-row x 16-column block of 16-bit elements and hands lane i of
// the group column i of the 4 rows.  Lane 4q+p of the group supplies the address of row q, columns 4p..4p+3
// (8 bytes, 8-byte aligned).  EXEC must be all ones.  Issued as inline asm:
// hipcc treats the ds_read_tr builtin as "may alias the LDS-DMA in flight" and
// drains vmcnt(0) in front of it, which serialises the global->LDS prefetch with the compute of every K-step;
// an asm read WITHOUT a "memory" clobber is invisible to that pass (with the clobber it drains just the same).  In exchange nothing waits for it: the caller issues a batch, then
// lds_tr_wait*() (s_waitcnt lgkmcnt(0) naming every destination, so no consumer or copy can move above it).
typedef unsigned long long sd_u64;
SD_DEV unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(SD_LDS const char*)p; }
SD_DEV void lds_tr16_pair_asm(sd_u64& lo, sd_u64& hi, unsigned a0, unsigned a1) {
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3" : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1));
}
SD_DEV bf16x8 cat8_u64(sd_u64 lo, sd_u64 hi) {
  typedef __attribute__((ext_vector_type(2))) sd_u64 u64x2;
  u64x2 v = {lo, hi};
  return __builtin_bit_cast(bf16x8, v);
}
SD_DEV void lds_tr_wait4(sd_u64 (&a)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
}
SD_DEV void lds_tr_wait8(sd_u64 (&a)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
}

SD_DEV f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
SD_DEV f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// Bijective XCD-aware block remap (8 XCDs; blocks b and b+8 share an XCD's L2): gives every XCD a
// contiguous run of logical tile ids.  Speed only, never correctness.
SD_DEV int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

#define SD_CHECK_LAUNCH() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)
