// Causal GQA flash attention (head_dim 128) forward + backward for gfx950, bf16 in/out, fp32 softmax.
//
// Replaces the flash_attn kernels the reference selects with attn_implementation="flash_attention_2"
// (train.py:160,177); the maths is HF eager_attention_forward (modeling_qwen3.py:185-207):
//     softmax(Q K^T d^-1/2 + causal/key-padding mask, fp32) V,  kv head j serves q heads jG..jG+G-1.
//
// All three kernels use v_mfma_f32_32x32x16_bf16 and ONE LDS image for every [rows][128] bf16 tile:
//     off(row, chunk16B) = 256*row + 16*(chunk ^ (((row&3)<<2) | ((row>>2)&3)))
// which is bank-conflict-free both for row reads (ds_read_b128, operand whose k index is d) and for
// transposed reads (ds_read_b64_tr_b16, operand whose k index is the row).  Tiles arrive by 16-byte
// LDS-DMA with the swizzle applied on the per-lane source address, double buffered.
//
// forward / dQ kernels: S^T = K Q^T, so a lane owns ONE query row (column l&31 of the 32x32 tile) and
//   the online softmax, the LSE / delta terms and the rescale are lane-local; P^T (dS^T) accumulator
//   registers feed the next MFMA as its B operand with no lane movement: O^T = V^T P^T, dQ^T = K^T dS^T.
// dK/dV kernel: S = Q K^T with the key on the lane; a wave owns 32 keys and keeps dK^T, dV^T in
//   registers over all query tiles and all G query heads of its kv head (no atomics, deterministic):
//   dV^T = dO^T P, dK^T = Q^T dS.
#include <type_traits>
#include "sd_common.cuh"
#include "../../include/sd_hip.h"
#include "sd_prof.h"
#include "sd_debug.h"
#include "sd_events.h"

namespace {

constexpr int D = 128;
constexpr int TILE = 64 * D * 2;  // one 64-row tile: 16 KiB
constexpr float NEG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;

SD_DEV int f_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// Stage 64 rows of a [*, 128]-column slice (row stride ld) into a 16 KiB LDS tile through a buffer descriptor
// (buffer_load_dwordx4 ... lds): per-lane offsets are computed once, a tile advance is one scalar offset, rows
// past the end of the (batch, head) slice read as zeros (they are masked later).
template <int NP = 4>  // 1 KiB pieces per wave: 4 when four waves stage a tile, 2 when all eight do
struct TileDma {
  int voff[NP];
  long row_bytes;
#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t rsrc;
#endif
  // g: first element of the slice (row 0 of this batch, this head); rows: rows in the slice
  SD_DEV void init(const bf16* g, long ld, int rows, int w, int lane) {
    row_bytes = ld * 2;
#if defined(__HIP_DEVICE_COMPILE__)
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, (int)(((long)(rows - 1) * ld + D) * 2), 0x00020000);
#endif
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = (w * NP + i) * 64 + lane;
      const int row = p >> 4, s = p & 15;
      voff[i] = (int)(((long)row * ld + ((s ^ f_swz(row)) * 8)) * 2);
    }
  }
  SD_DEV void issue(int row0, char* lds, int w) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const int soff = (int)(row0 * row_bytes);
#pragma unroll
    for (int i = 0; i < NP; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (SD_LDS void*)(lds + (w * NP + i) * 1024), 16, voff[i], soff, 0, 0);
#endif
  }
};

// operand whose k index is d: lane (r = l&31, h = l>>5) holds tile[row0 + r][16*st + 8h .. +7]
SD_DEV bf16x8 row_frag(const char* lds, int row0, int st, int lane) {
  const int row = row0 + (lane & 31);
  const int ch = 2 * st + (lane >> 5);
  return *(const bf16x8*)(lds + row * 256 + ((ch ^ f_swz(row)) << 4));
}

// Operand whose k index is the tile ROW (transposed read): lane (r = l&31 -> column db*32 + r, h = l>>5) holds
// tile[rb + 8(j>>2) + 4h + (j&3)][db*32 + r], j = 0..7  (rb = first row of this 16-row k-step) -- exactly the k
// order of accumulator registers 8s..8s+7 used as the other operand.  Two ds_read_b64_tr_b16 per fragment.
// The fragments for db = 0..3 at once, as asm reads (see lds_tr16_pair_asm: the builtin makes hipcc drain the K/V
// prefetch before every transposed read) followed by one lgkmcnt(0).
SD_DEV void tr_frag4(const char* lds, int rb, int lane, bf16x8 (&out)[4]) {
  const int gi = lane >> 4, i = lane & 15, q4 = i >> 2, pp = i & 3;
  const int rA = rb + 4 * (gi >> 1) + q4, rB = rA + 8;
  const int sA = f_swz(rA), sB = f_swz(rB);
  const unsigned base = lds_addr(lds) + 8 * (pp & 1);
  const unsigned aA = base + rA * 256, aB = base + rB * 256;
  const int chl = (gi & 1) * 2 + (pp >> 1);
  sd_u64 raw[8];
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    const int ch = db * 4 + chl;
    lds_tr16_pair_asm(raw[2 * db], raw[2 * db + 1], aA + ((ch ^ sA) << 4), aB + ((ch ^ sB) << 4));
  }
  lds_tr_wait8(raw);
#pragma unroll
  for (int db = 0; db < 4; ++db) out[db] = cat8_u64(raw[2 * db], raw[2 * db + 1]);
}

// the same reads without the wait: issue now, tr_finish4 later (the latency hides under whatever runs in between)
SD_DEV void tr_issue4(const char* lds, int rb, int lane, sd_u64 (&raw)[8]) {
  const int gi = lane >> 4, i = lane & 15, q4 = i >> 2, pp = i & 3;
  const int rA = rb + 4 * (gi >> 1) + q4, rB = rA + 8;
  const int sA = f_swz(rA), sB = f_swz(rB);
  const unsigned base = lds_addr(lds) + 8 * (pp & 1);
  const unsigned aA = base + rA * 256, aB = base + rB * 256;
  const int chl = (gi & 1) * 2 + (pp >> 1);
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    const int ch = db * 4 + chl;
    lds_tr16_pair_asm(raw[2 * db], raw[2 * db + 1], aA + ((ch ^ sA) << 4), aB + ((ch ^ sB) << 4));
  }
}
SD_DEV void tr_finish4(sd_u64 (&raw)[8], bf16x8 (&out)[4]) {
  lds_tr_wait8(raw);
#pragma unroll
  for (int db = 0; db < 4; ++db) out[db] = cat8_u64(raw[2 * db], raw[2 * db + 1]);
}

// max of a value and the one the lane 32 away holds (v_permlane32_swap: one instruction, no LDS round trip).  As asm:
// hipcc folds fmaxf(pr[0], pr[1]) of __builtin_amdgcn_permlane32_swap(u, u, ..) to pr[0] (it takes the two results of a
// swap of equal operands for equal), which silently leaves every row with the maximum of its first lane half only.
SD_DEV float halves_max(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));  // a = lo|lo, b = hi|hi
  return fmaxf(a, b);
}

SD_DEV bf16x8 acc_frag(const f32x16& p, int s) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (bf16)p[8 * s + j];
  return r;
}

// row index (within a 32x32 accumulator tile) of register `reg` for lane half h
SD_DEV int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// Epilogue store of a wave's 32 x 128 output tile held as four TRANSPOSED 32x32 accumulators (lane r = l&31 owns output
// ROW r; register e of block db is column db*32 + acc_row(e, h)).  Stored straight from the registers this is 16
// 8-byte stores per lane at a row stride -- every wave-instruction touches 32 different 256-byte rows, and the store
// ISSUE, not bandwidth, sets the time (~9k cycles per workgroup, MI355X_MICROARCH "attention epilogue store tail").
// Instead the tile goes through a wave-private 8 KiB LDS image in the [row][128] swizzled layout of every other tile
// and leaves as whole rows: 16 B per lane, four complete 256-byte rows per wave-instruction, 8 instructions.
SD_DEV void store_tile_rows(char* img, const f32x16 (&acc)[4], float mul, bf16* g_row0, long ld, int rows_valid, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const int swr = f_swz(r);
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      bf16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (bf16)(acc[db][4 * g4 + e] * mul);
      *(bf16x4*)(img + r * 256 + (((db * 4 + g4) ^ swr) << 4) + 8 * h) = v;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // one wave's LDS operations complete in order; keep the compiler's too
  const int ch = lane & 15;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = 4 * i + (lane >> 4);
    const bf16x8 v = *(const bf16x8*)(img + row * 256 + ((ch ^ f_swz(row)) << 4));
    if (row < rows_valid) *(bf16x8*)(g_row0 + (long)row * ld + ch * 8) = v;
  }
}

// Workgroup -> (pair, head, batch), XCD-aware.  The hardware deals workgroups to the 8 XCDs round-robin by linear id,
// and every XCD has its own 4 MiB L2.  The `gs` workgroups that stream the same tensors (forward / dQ: the tile pairs of
// the G query heads of one kv head; dK/dV: the key-block pairs of one kv head) would sit on `gs` different XCDs under the
// natural numbering, every L2 would pull every K/V stream (8.4 MB at B=4,T=512), and the first tile of a launch took
// 14 k cycles to arrive.  Here a group is dealt to ONE XCD: id -> (xcd = id % 8, slot = id / 8), group = xcd + 8 *
// (slot / gs), member = slot % gs (when the number of groups is a multiple of 8; the natural order otherwise).
struct AttnWg { int pair, member_head, hkv, b; };
SD_DEV AttnWg attn_wg(int id, int npairs, int heads_per_group, int Hkv, int B) {
  const int gs = npairs * heads_per_group, ng = Hkv * B;
  int g, mem;
  if ((ng & 7) == 0) {
    const int xcd = id & 7, slot = id >> 3;
    g = xcd + 8 * (slot / gs);
    mem = slot % gs;
  } else {
    g = id / gs;
    mem = id % gs;
  }
  AttnWg o;
  o.b = g / Hkv;
  o.hkv = g % Hkv;
  o.member_head = mem / npairs;
  o.pair = mem % npairs;
  return o;
}

#ifdef SD_STAMPS
// DIAGNOSTIC BUILD ONLY (make stamps): s_memtime at the phase boundaries of workgroup (0,0,0) of attn_fwd_kernel,
// [wave 0..7][point 0..255] (tests/bench_attn_stamps.py)
#define SD_ATT_STAMP_PARAM , unsigned long long* stamps
#define ATT_STAMP()                                                                                   \
  do {                                                                                                \
    if (stamping && lane == 0 && sidx < 256) stamps[w8 * 256 + sidx] = __builtin_amdgcn_s_memtime();   \
    ++sidx;                                                                                           \
  } while (0)
#define ATT_STAMP_AT(COND, SLOT)                                                                            \
  do {                                                                                                      \
    if (stamping && (COND) && lane == 0) stamps[w8 * 256 + (SLOT)] = __builtin_amdgcn_s_memtime();           \
  } while (0)
#else
#define SD_ATT_STAMP_PARAM
#define ATT_STAMP() do { } while (0)
#define ATT_STAMP_AT(COND, SLOT) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------ forward
// grid ceil(ceil(T/64)/2) * Hq * B (decoded by attn_wg), 512 threads.  Under the causal mask a 64-row query tile t sees t+1 K/V tiles, so
// equal row ranges are unequal work (at T = 512: 2, 4, 6, 8 tiles for the four 128-row tiles, and the chip waits for the
// heaviest).  A workgroup therefore takes a PAIR of 64-row query tiles, the j-th lightest and the j-th heaviest (every
// pair sees ceil(T/64)+1 tiles in total): waves 0-3 own the heavy tile, waves 4-7 the light one, and since waves w and
// w+4 share a SIMD every SIMD carries one wave of each.  Both tiles belong to the same (batch, head), so they read the
// same K/V stream: ONE ring streams the heavy tile's K/V tiles, the light tile's waves use the first of them and then
// only keep staging.  Inside a group of four: rb = w&1 is the 32-row block, half = (w>>1)&1 takes keys 32*half..+31 of
// EVERY K/V tile, so all eight waves walk one stream in lockstep through a FOUR-stage ring (4 x (K,V) = 128 KiB): three
// tiles are in flight while one is used.  (Until round 2 the halves took alternate tiles through a double buffer each:
// one tile ahead is less than the L2 latency with all 256 CUs pulling at once, and an iteration cost 3.2 us at T = 512
// against 0.85 us of MFMA time.)  With two waves per SIMD the DMA issue, LDS reads, MFMAs and softmax VALU of one wave
// overlap the other's.  The two partial (m, l, O) states of a row block are merged through LDS at the end (fixed order).
__global__ __launch_bounds__(512) void attn_fwd_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ Kp,
                                                       const bf16* __restrict__ Vp, bf16* __restrict__ O,
                                                       float* __restrict__ LSE, const int* __restrict__ kv_len, long ldq,
                                                       long ldk, long ldv, long ldo, int T, int Hq, int Hkv,
                                                       float scale SD_ATT_STAMP_PARAM) {
  __shared__ __attribute__((aligned(16))) char smem[8 * TILE];  // ring of 4 stages x (K,V)
  const int lane = lane_id(), w8 = wave_id_uniform();
  const int n64 = (T + 63) / 64;
  const AttnWg wg = attn_wg((int)blockIdx.x, (n64 + 1) / 2, Hq / Hkv, Hkv, (int)gridDim.x / (((n64 + 1) / 2) * Hq));
#ifdef SD_STAMPS
  const bool stamping = stamps && wg.pair == 0 && wg.member_head == 0 && wg.hkv == 0 && wg.b == 0;
  int sidx = 0;
#endif
  ATT_STAMP();  // 0: kernel entry
  const int grp = w8 >> 2, rb = w8 & 1, half = (w8 >> 1) & 1;
  const int w = grp * 2 + rb;  // 0..3 inside my half: staging share and merge slot
  // heavy / light 64-row query tile of this workgroup.  (Pairing ADJACENT tiles instead, which keeps both tiles' waves
  // busy for the whole stream, was measured for launches of several rounds of workgroups: slower at every shape, e.g.
  // B=16,T=512 62.5 vs 55.6 us, B=4,T=2048 134.5 vs 131.5 us -- two working waves per SIMD share one vector ALU.)
  const int tA = n64 - 1 - wg.pair, tB = wg.pair;
  const bool active = grp == 0 || tB != tA;      // odd tile count: the middle tile has no partner
  const int hkv = wg.hkv, b = wg.b;
  const int hq = hkv * (Hq / Hkv) + wg.member_head;
  const int q0w = (grp == 0 ? tA : tB) * 64 + 32 * rb;
  const int r = lane & 31, h = lane >> 5;
  const long tok0 = (long)b * T;
  const bf16* qb = Q + tok0 * ldq + hq * D;
  const bf16* kb = Kp + tok0 * ldk + hkv * D;
  const bf16* vb = Vp + tok0 * ldv + hkv * D;

  // The K/V stream starts before anything else: the first tiles take ~2 us to arrive (cold TLB / L2), and the Q loads,
  // the kv_len read and the rest of the set-up fit under that.
  const int kv_hi = min(tA * 64 + 64, T);  // the heavy tile's diagonal bounds the stream
  const int nkv = (kv_hi + 63) / 64;
  TileDma<2> kd, vd;  // all eight waves stage: two 1 KiB pieces of K and of V per wave and tile
  kd.init(kb, ldk, T, w8, lane);
  vd.init(vb, ldv, T, w8, lane);
  // ring of four (K,V) stages; tile i lives in stage i & 3.  Only tile 0 and the Q fragments go out now: the compiler
  // waits for ALL outstanding loads before the loop (Q lives in registers across it), so anything else issued here
  // would only delay the first tile.  Tiles 1..3 follow right after the first barrier, tile i+3 when tile i-1 retires.
  kd.issue(0, smem, w8);
  vd.issue(0, smem + TILE, w8);
  const int q = q0w + r;  // this lane's query row
  const int qc = q < T ? q : T - 1;
  bf16x8 qf[8];
#pragma unroll
  for (int st = 0; st < 8; ++st) qf[st] = *(const bf16x8*)(qb + (long)qc * ldq + 16 * st + 8 * h);
  ATT_STAMP();  // 1: prologue issued
  const int klen = kv_len ? max(1, min(kv_len[b], T)) : T;
  const int lim = min(q, klen - 1);  // keys > lim are masked for this row

  f32x16 o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
  float m = NEG, l = 0.f;
  const float c = scale * LOG2E;

  auto visible = [&](int i) { const int kv0 = i * 64 + 32 * half; return active && kv0 < kv_hi && kv0 <= q0w + 31; };
  // One tile = S^T (8 MFMAs, K fragments from LDS) -> online softmax (VALU) -> O^T += V^T P^T (8 MFMAs, transposed V
  // reads, requested before the softmax so that they are there when it ends).  The loop is VALU-bound, not latency- or
  // MFMA-bound (ISA count in round 2: 350 vector-ALU instructions per tile = ~1 600 issue cycles beside 512 cycles of
  // MFMA; two working waves per SIMD take twice as long), so everything that does not depend on the tile is computed
  // once: the per-lane LDS offsets of the K fragments and of the transposed V reads (a tile is then one add per read),
  // and the causal / padding mask sits behind a scalar branch that only the diagonal tiles take.
  unsigned kfo[8], vfo[2][8];
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const int row = 32 * half + (lane & 31), ch = 2 * st + (lane >> 5);
    kfo[st] = (unsigned)(row * 256 + ((ch ^ f_swz(row)) << 4));
  }
  {
    const int gi = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3;
    const int chl = (gi & 1) * 2 + (pp >> 1);
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const int rA = 32 * half + 16 * ss + 4 * (gi >> 1) + q4, rB = rA + 8;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const int ch = db * 4 + chl;
        vfo[ss][2 * db] = (unsigned)(8 * (pp & 1) + rA * 256 + ((ch ^ f_swz(rA)) << 4));
        vfo[ss][2 * db + 1] = (unsigned)(8 * (pp & 1) + rB * 256 + ((ch ^ f_swz(rB)) << 4));
      }
    }
  }
  const unsigned smem_a = lds_addr(smem);
  auto tile = [&](int i, int stage) {
    const char* ks = smem + stage * 2 * TILE;
    const unsigned vsa = smem_a + stage * 2 * TILE + TILE;
    f32x16 s1;
#pragma unroll
    for (int e = 0; e < 16; ++e) s1[e] = 0.f;
#pragma unroll
    for (int st = 0; st < 8; ++st) s1 = mfma32(*(const bf16x8*)(ks + kfo[st]), qf[st], s1);
    ATT_STAMP_AT(i == 3, 201);  // S^T MFMAs issued
    sd_u64 rv0[8], rv1[8];
#pragma unroll
    for (int db = 0; db < 4; ++db) lds_tr16_pair_asm(rv0[2 * db], rv0[2 * db + 1], vsa + vfo[0][2 * db], vsa + vfo[0][2 * db + 1]);
#pragma unroll
    for (int db = 0; db < 4; ++db) lds_tr16_pair_asm(rv1[2 * db], rv1[2 * db + 1], vsa + vfo[1][2 * db], vsa + vfo[1][2 * db + 1]);
    const int kv0 = i * 64 + 32 * half;
    // wave-uniform: my keys touch the diagonal / the padding (readfirstlane: a scalar branch, not 16 selects per tile)
    const bool need_mask = __builtin_amdgcn_readfirstlane((int)((kv0 + 31 > q0w) || (kv0 + 31 >= klen))) != 0;
    if (need_mask) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s1[e] = (kv0 + acc_row(e, h)) > lim ? NEG : s1[e];
    }
    float mx = fmaxf(fmaxf(s1[0], s1[1]), s1[2]);
#pragma unroll
    for (int e = 3; e < 15; e += 2) mx = fmaxf(fmaxf(mx, s1[e]), s1[e + 1]);
    mx = fmaxf(mx, s1[15]);
    ATT_STAMP_AT(i == 3, 202);  // lane maxima (S^T complete)
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    ATT_STAMP_AT(i == 3, 203);  // row maxima
    const float mn = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f((m - mn) * c);
    m = mn;
    const float mnc = -mn * c;
    float rs = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[e], c, mnc));
      s1[e] = p;
      rs += p;
    }
    l = l * alpha + rs;
    ATT_STAMP_AT(i == 3, 204);  // exponentials
    if (__any(alpha != 1.f)) {  // the running maximum moved for some row of this wave
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
    }
    ATT_STAMP_AT(i == 3, 205);  // rescaled
    bf16x8 vt[4];
    tr_finish4(rv0, vt);  // waits for rv1 too (LDS returns in order)
    ATT_STAMP_AT(i == 3, 206);  // V^T fragments here
    const bf16x8 p0 = acc_frag(s1, 0), p1 = acc_frag(s1, 1);
#pragma unroll
    for (int db = 0; db < 4; ++db) o[db] = mfma32(vt[db], p0, o[db]);
#pragma unroll
    for (int db = 0; db < 4; ++db) o[db] = mfma32(cat8_u64(rv1[2 * db], rv1[2 * db + 1]), p1, o[db]);
    ATT_STAMP_AT(i == 3, 207);  // P.V MFMAs issued
  };
  for (int i = 0; i < nkv; ++i) {
    // my share of tile i has landed; younger tiles (4 loads each) may still be in flight
    if (i == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (i + 2 < nkv) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (i + 1 < nkv) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATT_STAMP();  // 2+4i: my loads of tile i landed
    __builtin_amdgcn_s_barrier();  // everybody's share of tile i has landed; everybody is done reading tile i-1
    asm volatile("" ::: "memory");
    ATT_STAMP();  // 3+4i: barrier passed
    for (int t = (i == 0 ? 1 : i + 3); t <= i + 3; ++t)  // wave-uniform; 1, 2, 3 after the first barrier
      if (t < nkv) {
        char* nx = smem + (t & 3) * 2 * TILE;
        kd.issue(t * 64, nx, w8);
        vd.issue(t * 64, nx + TILE, w8);
      }
    ATT_STAMP();  // 4+4i: next tile issued
    if (visible(i)) {
      ATT_STAMP_AT(i == 3, 200);
      tile(i, i & 3);
    }
    ATT_STAMP();  // 5+4i: tile computed (issue side)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the tail tiles before the ring is reused
  __syncthreads();
  ATT_STAMP();  // 2+4nkv: loop left, ring free
  // merge the two halves: half 1 parks (m, l, O) in LDS, half 0 combines.  Layout [w][slot][lane] floats.
  float* xo = (float*)smem + (long)w * 66 * 64;
  if (half == 1) {
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) xo[(db * 16 + e) * 64 + lane] = o[db][e];
    xo[64 * 64 + lane] = m;
    xo[65 * 64 + lane] = l;
  }
  __syncthreads();
  if (half == 1) return;
  {
    const float m1 = xo[64 * 64 + lane], l1 = xo[65 * 64 + lane];
    const float mn = fmaxf(m, m1);
    const float a0 = __builtin_amdgcn_exp2f((m - mn) * c), a1 = __builtin_amdgcn_exp2f((m1 - mn) * c);
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[db][e] = o[db][e] * a0 + xo[(db * 16 + e) * 64 + lane] * a1;
    l = l * a0 + l1 * a1;
    m = mn;
  }
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.f / l;
  // every lane of a row has the same `inv` only after the two column halves are combined: lanes r and r+32 hold the
  // same row, and `l` was just summed over them, so scaling per lane before the transposing store is exact
  ATT_STAMP();  // 2+4nkv+1: merged
  if (!active) return;
  store_tile_rows((char*)xo, o, inv, O + (tok0 + q0w) * ldo + hq * D, ldo, T - q0w, lane);
  if (q < T && h == 0) LSE[((long)b * Hq + hq) * T + q] = m * scale + __logf(l);
  ATT_STAMP();  // stores issued
}

// The same decomposition, ring, merge and arithmetic (bit-identical outputs) with the K/V-tile loop SOFTWARE-PIPELINED
// inside the wave.  In attn_fwd_kernel a tile is one dependent chain -- K fragments -> 8 MFMAs -> maxima -> 16 v_exp ->
// bf16 P -> 8 MFMAs -- of ~2.1 k cycles for 512 cycles of matrix work, and on the heavy tile's SIMDs nothing else runs
// beside it (round-2 stamps, profiles/r02_attn_fwd_stamps.txt: 2.9 k cycles per tile with the DMA issue and the barrier).
// Here iteration i issues S^T(i+1) = K(i+1) Q^T FIRST and runs tile i's exponentials, row sums and bf16 conversion while the
// matrix pipe works on it, then issues O^T += V(i)^T P(i)^T and computes tile i+1's mask and row maxima beside those:
// both matrix products of an iteration have vector work of ANOTHER tile next to them.  The running-maximum update and
// the (rare) rescale of O^T sit at the top of the iteration, before anything is issued.  Tile i+1 has to be in LDS one
// iteration earlier than before, so the four-stage ring keeps two tiles in flight instead of three.
__global__ __launch_bounds__(512) void attn_fwd_pipe_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ Kp,
                                                            const bf16* __restrict__ Vp, bf16* __restrict__ O,
                                                            float* __restrict__ LSE, const int* __restrict__ kv_len,
                                                            long ldq, long ldk, long ldv, long ldo, int T, int Hq, int Hkv,
                                                            float scale SD_ATT_STAMP_PARAM) {
  __shared__ __attribute__((aligned(16))) char smem[8 * TILE];  // ring of 4 stages x (K,V)
  const int lane = lane_id(), w8 = wave_id_uniform();
  const int n64 = (T + 63) / 64;
  const AttnWg wg = attn_wg((int)blockIdx.x, (n64 + 1) / 2, Hq / Hkv, Hkv, (int)gridDim.x / (((n64 + 1) / 2) * Hq));
#ifdef SD_STAMPS
  const bool stamping = stamps && wg.pair == 0 && wg.member_head == 0 && wg.hkv == 0 && wg.b == 0;
  int sidx = 0;
#endif
  ATT_STAMP();  // 0: kernel entry
  const int grp = w8 >> 2, rb = w8 & 1, half = (w8 >> 1) & 1;
  const int w = grp * 2 + rb;
  const int tA = n64 - 1 - wg.pair, tB = wg.pair;
  const bool active = grp == 0 || tB != tA;
  const int hkv = wg.hkv, b = wg.b;
  const int hq = hkv * (Hq / Hkv) + wg.member_head;
  const int q0w = (grp == 0 ? tA : tB) * 64 + 32 * rb;
  const int r = lane & 31, h = lane >> 5;
  const long tok0 = (long)b * T;
  const bf16* qb = Q + tok0 * ldq + hq * D;
  const bf16* kb = Kp + tok0 * ldk + hkv * D;
  const bf16* vb = Vp + tok0 * ldv + hkv * D;
  const int kv_hi = min(tA * 64 + 64, T);
  const int nkv = (kv_hi + 63) / 64;
  TileDma<2> kd, vd;
  kd.init(kb, ldk, T, w8, lane);
  vd.init(vb, ldv, T, w8, lane);
  kd.issue(0, smem, w8);
  vd.issue(0, smem + TILE, w8);
  const int q = q0w + r;
  const int qc = q < T ? q : T - 1;
  bf16x8 qf[8];
#pragma unroll
  for (int st = 0; st < 8; ++st) qf[st] = *(const bf16x8*)(qb + (long)qc * ldq + 16 * st + 8 * h);
  const int klen = kv_len ? max(1, min(kv_len[b], T)) : T;
  const int lim = min(q, klen - 1);

  f32x16 o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
  float m = NEG, l = 0.f;
  const float c = scale * LOG2E;
  // tiles a wave computes: 0 .. nvis-1 (its keys of tile i start at 64 i + 32 half; at or below its rows' diagonal)
  int nvis = 0;
  if (active) {
    const int last = (q0w + 31 - 32 * half) >> 6;  // floor; negative when even tile 0 is above the diagonal
    nvis = (q0w + 31 - 32 * half) < 0 ? 0 : min(last + 1, (kv_hi - 32 * half + 63) >> 6);
  }
  // LDS offsets inside a tile.  The swizzle is an XOR on the 16-byte chunk index, so the eight K fragments of a lane are
  // ONE offset XOR (st << 5), and the transposed V reads of a 16-key step are two offsets XOR (db << 6): 5 registers
  // live across the loop instead of 24.
  unsigned kf0, vf0[2][2];
  {
    const int row = 32 * half + (lane & 31);
    kf0 = (unsigned)(row * 256 + (((lane >> 5) ^ f_swz(row)) << 4));
    const int gi = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3;
    const int chl = (gi & 1) * 2 + (pp >> 1);
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const int rA = 32 * half + 16 * ss + 4 * (gi >> 1) + q4, rB = rA + 8;
      vf0[ss][0] = (unsigned)(8 * (pp & 1) + rA * 256 + ((chl ^ f_swz(rA)) << 4));
      vf0[ss][1] = (unsigned)(8 * (pp & 1) + rB * 256 + ((chl ^ f_swz(rB)) << 4));
    }
  }
  const unsigned smem_a = lds_addr(smem);
  // S^T of one tile (this wave's 32 keys x its 32 rows): eight chained MFMAs on K fragments read from `stage`
  auto s_tile = [&](int stage, f32x16& s) __attribute__((always_inline)) {
    const char* ks = smem + stage * 2 * TILE;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
    for (int st = 0; st < 8; ++st) s = mfma32(*(const bf16x8*)(ks + (kf0 ^ (unsigned)(st << 5))), qf[st], s);
  };
  // causal / padding mask (diagonal tiles only, behind a scalar branch) and the row maxima of a fresh S^T tile
  auto mask_max = [&](int t, f32x16& s) __attribute__((always_inline)) -> float {
    const int kv0 = t * 64 + 32 * half;
    const bool need_mask = __builtin_amdgcn_readfirstlane((int)((kv0 + 31 > q0w) || (kv0 + 31 >= klen))) != 0;
    if (need_mask) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = (kv0 + acc_row(e, h)) > lim ? NEG : s[e];
    }
    float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
    for (int e = 3; e < 15; e += 2) mx = fmaxf(fmaxf(mx, s[e]), s[e + 1]);
    mx = fmaxf(mx, s[15]);
    return halves_max(mx);
  };

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // Q fragments and my share of tile 0
  // hipcc does not see that wait: without a use of the Q registers HERE it would wait for their loads at their first
  // use inside the loop, with vmcnt(7..0) -- which by then drains the K/V tiles in flight
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    u32x4 t4 = __builtin_bit_cast(u32x4, qf[st]);
    asm volatile("" : "+v"(t4));
    qf[st] = __builtin_bit_cast(bf16x8, t4);
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  for (int t = 1; t <= 2; ++t)
    if (t < nkv) {
      char* nx = smem + (t & 3) * 2 * TILE;
      kd.issue(t * 64, nx, w8);
      vd.issue(t * 64, nx + TILE, w8);
    }
  f32x16 sc, sn;
  float mx = NEG;
  if (nvis > 0) {
    s_tile(0, sc);
    mx = mask_max(0, sc);
  }
  ATT_STAMP();  // 1: prologue done (tile 0 landed, S^T(0) issued)
  for (int i = 0; i < nkv; ++i) {
    // tile i+1 (whose K fragments this iteration reads) has landed; tile i+2 (4 loads of mine) may still be in flight
    if (i + 2 < nkv) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATT_STAMP();  // 2+4i: my loads of tile i+1 landed
    __builtin_amdgcn_s_barrier();  // everybody's share of tile i+1 is there; everybody is done with tile i-1
    asm volatile("" ::: "memory");
    ATT_STAMP();  // 3+4i: barrier passed
    const bool comp = i < nvis;  // wave-uniform
    // All LDS reads of the iteration go out first, as asm (hipcc keeps only two K fragments in flight and then paces
    // the S^T chain by the LDS latency: ~100 cycles per MFMA instead of 32): the 8 K fragments of tile i+1, then the
    // transposed V reads of tile i's keys 0-15 of this wave; keys 16-31 follow once the K fragments are back (the
    // LDS counter holds 15).  ONE code path: S^T(i+1) is also issued after the wave's last tile (on whatever the next
    // stage holds; the result is dropped) -- a second path without it made hipcc copy the 64 O^T registers around the
    // join every iteration.
    const unsigned ksa = smem_a + ((i + 1) & 3) * 2 * TILE + kf0;
    const unsigned vsa = smem_a + (i & 3) * 2 * TILE + TILE;
    const unsigned va[2][2] = {{vsa + vf0[0][0], vsa + vf0[0][1]}, {vsa + vf0[1][0], vsa + vf0[1][1]}};
    u32x4 kf[8];
    sd_u64 rv0[8], rv1[8];
    if (comp) {
#pragma unroll
      for (int st = 0; st < 8; ++st) asm volatile("ds_read_b128 %0, %1" : "=&v"(kf[st]) : "v"(ksa ^ (unsigned)(st << 5)));
#pragma unroll
      for (int db = 0; db < 4; ++db) lds_tr16_pair_asm(rv0[2 * db], rv0[2 * db + 1], va[0][0] ^ (unsigned)(db << 6), va[0][1] ^ (unsigned)(db << 6));
    }
    // the next tile's DMA goes out while those reads are in flight (~80 cycles of issue per piece)
    if (i + 3 < nkv) {
      char* nx = smem + ((i + 3) & 3) * 2 * TILE;
      kd.issue((i + 3) * 64, nx, w8);
      vd.issue((i + 3) * 64, nx + TILE, w8);
    }
    ATT_STAMP();  // 4+4i: reads and tile i+3 issued
    if (!comp) {
      ATT_STAMP();
      continue;
    }
    const float mn = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f((m - mn) * c);
    m = mn;
    const float mnc = -mn * c;
    if (__any(alpha != 1.f)) {  // the running maximum moved for some row of this wave
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
    }
    asm volatile("s_waitcnt lgkmcnt(8)"
                 : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(kf[4]), "+v"(kf[5]), "+v"(kf[6]), "+v"(kf[7]));
#pragma unroll
    for (int db = 0; db < 4; ++db) lds_tr16_pair_asm(rv1[2 * db], rv1[2 * db + 1], va[1][0] ^ (unsigned)(db << 6), va[1][1] ^ (unsigned)(db << 6));
#pragma unroll
    for (int e = 0; e < 16; ++e) sn[e] = 0.f;
    // every MFMA of the S^T(i+1) chain with its share of tile i's vector work behind it, fenced: two exponentials
    // (16 of the gap's 32 cycles), their two multiply-adds and two row-sum adds
    float rs = 0.f;
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      sn = mfma32(__builtin_bit_cast(bf16x8, kf[st]), qf[st], sn);
#pragma unroll
      for (int e = 2 * st; e < 2 * st + 2; ++e) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[e], c, mnc));
        sc[e] = p;
        rs += p;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    l = l * alpha + rs;
    const bf16x8 p0 = acc_frag(sc, 0), p1 = acc_frag(sc, 1);
    asm volatile("s_waitcnt lgkmcnt(8)"
                 : "+v"(rv0[0]), "+v"(rv0[1]), "+v"(rv0[2]), "+v"(rv0[3]), "+v"(rv0[4]), "+v"(rv0[5]), "+v"(rv0[6]), "+v"(rv0[7]));
#pragma unroll
    for (int db = 0; db < 4; ++db) o[db] = mfma32(cat8_u64(rv0[2 * db], rv0[2 * db + 1]), p0, o[db]);
    lds_tr_wait8(rv1);
#pragma unroll
    for (int db = 0; db < 4; ++db) o[db] = mfma32(cat8_u64(rv1[2 * db], rv1[2 * db + 1]), p1, o[db]);
    mx = mask_max(i + 1, sn);
    sc = sn;
    ATT_STAMP();  // 5+4i: tile i computed (issue side)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ATT_STAMP();  // 2+4nkv: loop left, ring free
  float* xo = (float*)smem + (long)w * 66 * 64;
  if (half == 1) {
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) xo[(db * 16 + e) * 64 + lane] = o[db][e];
    xo[64 * 64 + lane] = m;
    xo[65 * 64 + lane] = l;
  }
  __syncthreads();
  if (half == 1) return;
  {
    const float m1 = xo[64 * 64 + lane], l1 = xo[65 * 64 + lane];
    const float mn = fmaxf(m, m1);
    const float a0 = __builtin_amdgcn_exp2f((m - mn) * c), a1 = __builtin_amdgcn_exp2f((m1 - mn) * c);
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[db][e] = o[db][e] * a0 + xo[(db * 16 + e) * 64 + lane] * a1;
    l = l * a0 + l1 * a1;
    m = mn;
  }
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.f / l;
  ATT_STAMP();  // merged
  if (!active) return;
  store_tile_rows((char*)xo, o, inv, O + (tok0 + q0w) * ldo + hq * D, ldo, T - q0w, lane);
  if (q < T && h == 0) LSE[((long)b * Hq + hq) * T + q] = m * scale + __logf(l);
  ATT_STAMP();  // stores issued
}

// -------------------------------------------------------------------------- delta = rowsum(dO * O)
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16* __restrict__ dO, const bf16* __restrict__ O,
                                                         float* __restrict__ delta, long ldo, int T, int Hq, long total) {
  const int lane = lane_id();
  const long idx = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);  // (token, head)
  const int j = lane & 15;
  const bool ok = idx < total;
  const long id = ok ? idx : total - 1;
  const long tok = id / Hq;
  const int hh = (int)(id % Hq);
  bf16x8 a = *(const bf16x8*)(dO + tok * ldo + hh * D + j * 8);
  bf16x8 bb = *(const bf16x8*)(O + tok * ldo + hh * D + j * 8);
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += (float)a[e] * (float)bb[e];
  s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
  if (ok && j == 0) {
    const long bidx = tok / T;
    const int t = (int)(tok % T);
    delta[(bidx * Hq + hh) * T + t] = s;
  }
}

// ----------------------------------------------------------------------------------------------- dQ
// Same decomposition as forward: a workgroup takes the j-th heaviest and the j-th lightest 64-row query tile of one
// (batch, head) -- waves 0-3 the heavy one, waves 4-7 the light one, one of each per SIMD -- over ONE K/V stream; inside
// a group rb = w&1 is the 32-row block and half = (w>>1)&1 splits the K/V tiles (t = 2i + half) through its own double
// buffer; the two partial dQ^T of a row block are added through LDS at the end in a fixed order.
// dQ^T[d][q] = scale * sum_key K^T[d][key] dS^T[key][q].
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ Kp,
                                                          const bf16* __restrict__ Vp, const bf16* __restrict__ dO,
                                                          const float* __restrict__ LSE, const float* __restrict__ delta,
                                                          bf16* __restrict__ dQ, const int* __restrict__ kv_len, long ldq,
                                                          long ldk, long ldv, long ldo, long lddq, int T, int Hq, int Hkv,
                                                          float scale) {
  __shared__ __attribute__((aligned(16))) char smem[8 * TILE];  // 2 halves x 2 stages x (K,V)
  const int lane = lane_id(), w8 = wave_id_uniform();
  const int grp = w8 >> 2, rb = w8 & 1, half = (w8 >> 1) & 1;
  const int w = grp * 2 + rb;  // 0..3 inside my half: staging share and merge slot
  const int n64 = (T + 63) / 64;
  const AttnWg wg = attn_wg((int)blockIdx.x, (n64 + 1) / 2, Hq / Hkv, Hkv, (int)gridDim.x / (((n64 + 1) / 2) * Hq));
  const int tA = n64 - 1 - wg.pair, tB = wg.pair;
  const bool active = grp == 0 || tB != tA;
  const int hkv = wg.hkv, b = wg.b;
  const int hq = hkv * (Hq / Hkv) + wg.member_head;
  const int q0w = (grp == 0 ? tA : tB) * 64 + 32 * rb;
  const int r = lane & 31, h = lane >> 5;
  const int klen = kv_len ? max(1, min(kv_len[b], T)) : T;
  const long tok0 = (long)b * T;
  const bf16* qb = Q + tok0 * ldq + hq * D;
  const bf16* kb = Kp + tok0 * ldk + hkv * D;
  const bf16* vb = Vp + tok0 * ldv + hkv * D;
  const bf16* dob = dO + tok0 * ldo + hq * D;
  const int q = q0w + r;
  const int qc = q < T ? q : T - 1;
  bf16x8 qf[8], dof[8];
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    qf[st] = *(const bf16x8*)(qb + (long)qc * ldq + 16 * st + 8 * h);
    dof[st] = *(const bf16x8*)(dob + (long)qc * ldo + 16 * st + 8 * h);
  }
  const int lim = min(q, klen - 1);
  const float lse2 = LSE[((long)b * Hq + hq) * T + qc] * LOG2E;
  const float dl = delta[((long)b * Hq + hq) * T + qc];
  const float c = scale * LOG2E;
  f32x16 acc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[db][e] = 0.f;

  const int kv_hi = min(tA * 64 + 64, T);
  const int nkv = (kv_hi + 63) / 64;
  const int nit = (nkv + 1) >> 1;  // trips of both halves; half 1 idles through the last one when nkv is odd
  TileDma<> kd, vd;
  kd.init(kb, ldk, T, w, lane);
  vd.init(vb, ldv, T, w, lane);
  char* ring = smem + half * 4 * TILE;
  kd.issue(half * 64, ring, w);
  vd.issue(half * 64, ring + TILE, w);
  int cur_i = 0;
  for (int i = 0; i < nit; ++i) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // my half's tile i has landed; everybody is done reading the other stage
    asm volatile("" ::: "memory");
    if ((2 * (i + 1) + half) * 64 < kv_hi) {  // wave-uniform; a tile nobody will read is not fetched (and not waited for)
      char* nx = ring + (cur_i ^ 1) * 2 * TILE;
      kd.issue((2 * (i + 1) + half) * 64, nx, w);
      vd.issue((2 * (i + 1) + half) * 64, nx + TILE, w);
    }
    const char* ks = ring + cur_i * 2 * TILE;
    const char* vs = ks + TILE;
    cur_i ^= 1;
    const int kv0 = (2 * i + half) * 64;
    if (active && kv0 < kv_hi && kv0 <= q0w + 31) {
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2) {
        f32x16 s, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
        for (int st = 0; st < 8; ++st) {
          s = mfma32(row_frag(ks, kb2 * 32, st, lane), qf[st], s);     // S^T  = K Q^T
          dp = mfma32(row_frag(vs, kb2 * 32, st, lane), dof[st], dp);  // dP^T = V dO^T
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kv0 + kb2 * 32 + acc_row(e, h);
          const float p = key > lim ? 0.f : __builtin_amdgcn_exp2f(s[e] * c - lse2);
          s[e] = p * (dp[e] - dl);  // dS^T (without the d^-1/2 factor)
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const bf16x8 df = acc_frag(s, ss);
          bf16x8 kt4[4];
          tr_frag4(ks, kb2 * 32 + 16 * ss, lane, kt4);
#pragma unroll
          for (int db = 0; db < 4; ++db) acc[db] = mfma32(kt4[db], df, acc[db]);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the tail tiles before the ring is reused
  __syncthreads();
  float* xo = (float*)smem + (long)w * 64 * 64;  // [w][slot][lane] floats
  if (half == 1) {
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) xo[(db * 16 + e) * 64 + lane] = acc[db][e];
  }
  __syncthreads();
  if (half == 1 || !active) return;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[db][e] += xo[(db * 16 + e) * 64 + lane];
  store_tile_rows((char*)xo, acc, scale, dQ + (tok0 + q0w) * lddq + hq * D, lddq, T - q0w, lane);
}

// -------------------------------------------------------------------------------------------- dK, dV
// grid ceil(ceil(T/64)/2) * Hkv * B (decoded by attn_wg), 512 threads = 8 waves, two per SIMD.  Key block kb (64 keys) is seen by the query
// tiles kb .. ceil(T/64)-1, so the first blocks carry most of the work; a workgroup takes the j-th heaviest and the
// j-th lightest block (every pair: ceil(T/64)+1 block-tile visits per query head): waves 0-3 the heavy one, waves 4-7
// the light one, one of each per SIMD.  Inside a group sub = w&1 is the 32-key sub-block a wave owns (the lane owns one
// key: S = Q K^T with the key on the lane) and rh = (w>>1)&1 the 32-row half of every 64-row (Q, dO) tile it takes.
// The block loops over the G query heads of the group and over the query tiles at or below the heavy block's
// diagonal (the light block's waves skip the tiles above theirs); a wave keeps dK^T, dV^T of its keys in registers
// (dV^T = dO^T P, dK^T = Q^T dS), and the two row halves are added through LDS at the end in a fixed order (no atomics,
// deterministic).  With ONE wave per SIMD a wave ran its DMA issue, LDS reads, 64 MFMAs and the exp/mask VALU of a
// tile strictly one after the other; two independent waves per SIMD overlap those phases.  To fit two waves into the
// SIMD's 512 registers the two blocks' K and V (128 keys) live in LDS instead of registers (64 KiB), beside a 2-stage
// (Q, dO) ring (64 KiB): waves 0-3 stage K and the Q tiles, waves 4-7 stage V and the dO tiles.
__global__ __launch_bounds__(512) void attn_bwd_dkv_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ Kp,
                                                           const bf16* __restrict__ Vp, const bf16* __restrict__ dO,
                                                           const float* __restrict__ LSE, const float* __restrict__ delta,
                                                           bf16* __restrict__ dK, bf16* __restrict__ dV,
                                                           const int* __restrict__ kv_len, long ldq, long ldk, long ldv,
                                                           long ldo, long lddk, long lddv, int T, int Hq, int Hkv,
                                                           float scale) {
  // [K: heavy block 64 rows | light block 64 rows][V: same][stage 0: Q, dO | stage 1: Q, dO | stat[2 stages][lse2|delta][64 q rows]]
  // (ONE __shared__ object: a second one beside an LDS-DMA target makes hipcc drain vmcnt before LDS reads)
  __shared__ __attribute__((aligned(16))) char smem[8 * TILE + 1024];
  const int lane = lane_id(), w8 = wave_id_uniform();
  const int grp = w8 >> 2, sub = w8 & 1, rh = (w8 >> 1) & 1;  // compute role
  const int sw = w8 & 3;                                       // staging share inside my group of four
  const int G = Hq / Hkv;
  const int n64 = (T + 63) / 64;
  const AttnWg wg = attn_wg((int)blockIdx.x, (n64 + 1) / 2, 1, Hkv, (int)gridDim.x / (((n64 + 1) / 2) * Hkv));
  const int hkv = wg.hkv, b = wg.b;
  const int kbA = wg.pair, kbB = n64 - 1 - wg.pair;  // heavy / light key block
  const bool active = grp == 0 || kbB != kbA;                       // odd block count: the middle block has no partner
  const int k0w = (grp == 0 ? kbA : kbB) * 64 + 32 * sub;
  const int r = lane & 31, h = lane >> 5;
  const int klen = kv_len ? max(1, min(kv_len[b], T)) : T;
  const long tok0 = (long)b * T;
  const int key = k0w + r;  // this lane's key (column of S)
  const bool key_ok = key < klen;  // padded / out-of-range keys receive no probability
  const float c = scale * LOG2E;
  char* const ksm = smem;
  char* const vsm = smem + 2 * TILE;
  char* const ring = smem + 4 * TILE;
  float* const stat = (float*)(smem + 8 * TILE);  // [stage][which][64]
  {
    TileDma<> kvd;
    kvd.init((grp ? Vp + tok0 * ldv : Kp + tok0 * ldk) + hkv * D, grp ? ldv : ldk, T, sw, lane);
    char* dst = grp ? vsm : ksm;
    kvd.issue(kbA * 64, dst, sw);
    kvd.issue(kbB * 64, dst + TILE, sw);
  }
  f32x16 dk[4], dv[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) { dk[db][e] = 0.f; dv[db][e] = 0.f; }

  const int qt0 = kbA;                 // first 64-row query tile that can see a key of the heavy block
  const int per_head = n64 - qt0;
  const int nit = G * per_head;
  const int st_i = threadIdx.x & 63, st_which = (threadIdx.x >> 6) & 1;
  // LSE / delta of the 64 query rows of tile `it` (threads 0..127), loaded an iteration before they are written to LDS
  auto load_stat = [&](int it) -> float {
    if (threadIdx.x >= 128) return 0.f;
    const int itc = it < nit ? it : nit - 1;
    const int hq = hkv * G + itc / per_head;
    int qq = (qt0 + itc % per_head) * 64 + st_i;
    qq = qq < T ? qq : T - 1;
    const long o = ((long)b * Hq + hq) * T + qq;
    return st_which ? delta[o] : LSE[o] * LOG2E;
  };
  // one tile = Q rows (waves 0-3) + dO rows (waves 4-7) of query head g, rows qt*64 .. +63
  auto stage_it = [&](int it, int buf) {
    const int g = it / per_head, qt = qt0 + it % per_head;
    const int hq = hkv * G + g;
    TileDma<> d;  // the query head changes with `it`: the descriptor is rebuilt (scalar work only)
    d.init((grp ? dO + tok0 * ldo : Q + tok0 * ldq) + hq * D, grp ? ldo : ldq, T, sw, lane);
    d.issue(qt * 64, ring + buf * 2 * TILE + grp * TILE, sw);
  };
  {
    const float sv0 = load_stat(0);
    stage_it(0, 0);
    if (threadIdx.x < 128) stat[st_which * 64 + st_i] = sv0;
  }
  float pend_sv = load_stat(1);
  for (int it = 0; it < nit; ++it) {
    const int buf = it & 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // tile `it` (and K, V) landed; everybody has finished reading the other stage
    asm volatile("" ::: "memory");
    if (threadIdx.x < 128) stat[(buf ^ 1) * 128 + st_which * 64 + st_i] = pend_sv;
    pend_sv = load_stat(it + 2);
    if (it + 1 < nit) stage_it(it + 1, buf ^ 1);
    const char* qs = ring + buf * 2 * TILE;
    const char* dos = qs + TILE;
    const int qb0 = (qt0 + it % per_head) * 64 + 32 * rh;
    if (!active || qb0 + 31 < k0w) continue;  // wave-uniform: this wave's 32 query rows are all above its keys
    f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      s = mfma32(row_frag(qs, 32 * rh, st, lane), row_frag(ksm, 64 * grp + 32 * sub, st, lane), s);     // S  = Q K^T
      dp = mfma32(row_frag(dos, 32 * rh, st, lane), row_frag(vsm, 64 * grp + 32 * sub, st, lane), dp);  // dP = dO V^T
    }
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4 l2 = *(const f32x4*)&stat[buf * 128 + 32 * rh + 8 * g4 + 4 * h];
      const f32x4 dl = *(const f32x4*)&stat[buf * 128 + 64 + 32 * rh + 8 * g4 + 4 * h];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int qq = qb0 + 8 * g4 + 4 * h + e;
        const bool vis = key_ok && key <= qq && qq < T;
        const float p = vis ? __builtin_amdgcn_exp2f(s[4 * g4 + e] * c - l2[e]) : 0.f;
        s[4 * g4 + e] = p;
        dp[4 * g4 + e] = p * (dp[4 * g4 + e] - dl[e]);  // dS (without the d^-1/2 factor)
      }
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      bf16x8 t4[4];
      const bf16x8 pf = acc_frag(s, ss);
      tr_frag4(dos, 32 * rh + 16 * ss, lane, t4);
#pragma unroll
      for (int db = 0; db < 4; ++db) dv[db] = mfma32(t4[db], pf, dv[db]);  // dV^T += dO^T P
      const bf16x8 df = acc_frag(dp, ss);
      tr_frag4(qs, 32 * rh + 16 * ss, lane, t4);
#pragma unroll
      for (int db = 0; db < 4; ++db) dk[db] = mfma32(t4[db], df, dk[db]);  // dK^T += Q^T dS
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // drain before LDS is reused
  __syncthreads();
  // merge: row half 1 parks dK^T then dV^T of its keys in LDS ([slot][reg][lane] floats, 16 KiB per wave and round)
  float* xo = (float*)smem + (long)(grp * 2 + sub) * 64 * 64;
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    f32x16* acc = round ? dv : dk;
    if (rh == 1) {
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) xo[(db * 16 + e) * 64 + lane] = acc[db][e];
    }
    __syncthreads();
    if (rh == 0) {
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[db][e] += xo[(db * 16 + e) * 64 + lane];
    }
    __syncthreads();
  }
  if (rh == 0 && active) {
    store_tile_rows((char*)xo, dk, scale, dK + (tok0 + k0w) * lddk + hkv * D, lddk, T - k0w, lane);
    store_tile_rows((char*)xo + 8192, dv, 1.f, dV + (tok0 + k0w) * lddv + hkv * D, lddv, T - k0w, lane);
  }
}

int check_common(int B, int T, int Hq, int Hkv, long ldq, long ldk, long ldv, long ldo) {
  if (B <= 0 || T <= 0 || Hq <= 0 || Hkv <= 0 || (Hq % Hkv)) return SD_ERR_SHAPE;
  if ((ldq | ldk | ldv | ldo) & 7) return SD_ERR_ALIGN;
  return 0;
}

}  // namespace

#ifdef SD_STAMPS
static unsigned long long* g_attn_stamps = nullptr;
extern "C" void sd_debug_attn_stamp_buffer(void* p) { g_attn_stamps = (unsigned long long*)p; }
#define SD_ATT_STAMP_ARG , g_attn_stamps
#else
#define SD_ATT_STAMP_ARG
#endif

// bit 0: forward without the in-wave software pipeline (attn_fwd_kernel) at any T; bit 1: with it at any T (default:
// from T = 1024); tests and A/B measurements
// (g_sd_debug.attn_variant, include/sd_hip_debug.h key "attn.variant")

extern "C" int sd_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int32_t* kv_len,
                           int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int B, int T, int Hq, int Hkv,
                           int head_dim, float scale, void* stream) {
  if (head_dim != D) return SD_ERR_UNSUPPORTED;
  if (int e = check_common(B, T, Hq, Hkv, ldq, ldk, ldv, ldo)) return e;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) return SD_ERR_ALIGN;
  SdProfScope prof(SD_K_ATTN_FWD, 2.0 * B * Hq * (double)T * T * D, (hipStream_t)stream);  // 2 products, causal half
  const dim3 grid((((T + 63) / 64 + 1) / 2) * Hq * B);
  // Measured (tests/bench_attn_pipe.py, MI355X, 16/8 heads): the pipelined kernel is 3-5 % faster from T = 1024 (21.8 vs
  // 22.2 us at B=2, 32.9 vs 34.6 us at B=1 T=2048, 119 vs 123 us at B=4 T=2048) and ties or loses 0.1-0.5 us below
  // (its prologue computes S^T(0) before the loop and needs tile 1 for the first iteration), so short streams keep the
  // classic kernel.  bit 1 of the variant forces the pipelined one at any T (tests).
  const bool pipe = (g_sd_debug.attn_variant & 2) || (!(g_sd_debug.attn_variant & 1) && T >= 1024);
  if (pipe) {
    hipLaunchKernelGGL(attn_fwd_pipe_kernel, grid, dim3(512), 0, (hipStream_t)stream, (const bf16*)q, (const bf16*)k,
                       (const bf16*)v, (bf16*)o, lse, kv_len, ldq, ldk, ldv, ldo, T, Hq, Hkv, scale SD_ATT_STAMP_ARG);
    SD_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(512), 0, (hipStream_t)stream, (const bf16*)q,
                     (const bf16*)k, (const bf16*)v, (bf16*)o, lse, kv_len, ldq, ldk, ldv, ldo, T, Hq, Hkv, scale SD_ATT_STAMP_ARG);
  SD_CHECK_LAUNCH();
  return 0;
}

extern "C" int sd_attn_bwd2(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                            float* delta, void* dq, void* dk, void* dv, const int32_t* kv_len, int64_t ldq, int64_t ldk,
                            int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv, int B, int T, int Hq,
                            int Hkv, int head_dim, float scale, void* side_stream, void* stream) {
  if (head_dim != D) return SD_ERR_UNSUPPORTED;
  if (int e = check_common(B, T, Hq, Hkv, ldq, ldk, ldv, ldo)) return e;
  if ((lddq | lddk | lddv) & 7) return SD_ERR_ALIGN;
  hipStream_t st = (hipStream_t)stream, s2 = (hipStream_t)side_stream;
  SdEventLease lease;  // this call's events (per call, per device)
  if (s2 && !(lease.set = sd_lease_events())) return SD_ERR_WORKSPACE;
  hipEvent_t* g_attn_ev = lease.set ? lease.set->ev : nullptr;
  const long total = (long)B * T * Hq;
  if (o) {  // o == NULL: `delta` already holds rowsum(dO * O) (sd_gemm_odx_delta)
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, st, (const bf16*)d_o,
                       (const bf16*)o, delta, ldo, T, Hq, total);
    SD_CHECK_LAUNCH();
  }
  hipStream_t sq = s2 ? s2 : st;  // stream of the dQ kernel
  if (s2) {
    if (hipEventRecord(g_attn_ev[0], st) != hipSuccess || hipStreamWaitEvent(s2, g_attn_ev[0], 0) != hipSuccess)
      return SD_ERR_WORKSPACE;
  }
  {
    SdProfScope prof2(SD_K_ATTN_BWD_DQ, 3.0 * B * Hq * (double)T * T * D, sq);  // S, dP (recomputed), dQ
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((((T + 63) / 64 + 1) / 2) * Hq * B), dim3(512), 0, sq, (const bf16*)q, (const bf16*)k,
                       (const bf16*)v, (const bf16*)d_o, lse, (const float*)delta, (bf16*)dq, kv_len, ldq, ldk, ldv, ldo,
                       lddq, T, Hq, Hkv, scale);
  }
  SD_CHECK_LAUNCH();
  {
    SdProfScope prof(SD_K_ATTN_BWD_DKV, 4.0 * B * Hq * (double)T * T * D, st);  // S, dP, dV, dK
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((((T + 63) / 64 + 1) / 2) * Hkv * B), dim3(512), 0, st, (const bf16*)q,
                       (const bf16*)k, (const bf16*)v, (const bf16*)d_o, lse, (const float*)delta, (bf16*)dk, (bf16*)dv,
                       kv_len, ldq, ldk, ldv, ldo, lddk, lddv, T, Hq, Hkv, scale);
  }
  SD_CHECK_LAUNCH();
  if (s2) {
    if (hipEventRecord(g_attn_ev[1], s2) != hipSuccess || hipStreamWaitEvent(st, g_attn_ev[1], 0) != hipSuccess)
      return SD_ERR_WORKSPACE;
  }
  return 0;
}

extern "C" int sd_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                           float* delta, void* dq, void* dk, void* dv, const int32_t* kv_len, int64_t ldq, int64_t ldk,
                           int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv, int B, int T, int Hq,
                           int Hkv, int head_dim, float scale, void* stream) {
  return sd_attn_bwd2(q, k, v, o, d_o, lse, delta, dq, dk, dv, kv_len, ldq, ldk, ldv, ldo, lddq, lddk, lddv, B, T, Hq,
                      Hkv, head_dim, scale, nullptr, stream);
}
