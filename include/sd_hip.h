/* sd_hip.h -- C ABI of libsd_hip.so: the gfx950 (MI355X / CDNA4) kernels and step runner of the
 * Stage-2 distillation hot path of indiejoseph/speech-distill.
 *
 * The reference is pure Python and binds nothing native for this path; what it CALLS there are
 * third-party CUDA kernels (flash_attn, cuBLAS, ATen).  Each entry point below cites the reference
 * (or third-party HF) code whose arithmetic it replaces.  "HF:" = transformers/models/qwen3/
 * modeling_qwen3.py, the decoder the reference instantiates at train.py:155-178.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; bf16 tensors are row-major,
 *     16-byte aligned, leading dimensions are in ELEMENTS and multiples of 8;
 *   - `stream` is a hipStream_t (torch's current stream); launchers never allocate device memory and never
 *     synchronise; scratch comes from the caller (see *_workspace_bytes).  Process state reachable from THIS header is
 *     limited to: (1) a pool of timing-disabled hipEvent_t sets, leased per sd_qwen3_backward* / sd_attn_bwd2 call and per
 *     device (two concurrent backward calls never share an event); (2) the sd_prof_* accumulators (only between
 *     sd_prof_begin/end).  The library never reads the environment.  Measurement and test switches (forced kernel variants,
 *     A/B thresholds, the CU budget of a multi-GPU run) are a separate interface, include/sd_hip_debug.h;
 *   - return value: SD_OK (0), a negative SD_ERR_* code, or a positive hipError_t from the launch.
 */
#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SD_OK 0
#define SD_ERR_SHAPE (-1)
#define SD_ERR_ALIGN (-2)
#define SD_ERR_UNSUPPORTED (-3)
#define SD_ERR_NO_TEACHER (-4) /* distillation_loss.py:120 ValueError("Either teacher_logits or top_k must be provided") */
#define SD_ERR_WORKSPACE (-5)

#define SD_DTYPE_BF16 0
#define SD_DTYPE_F32 1

int sd_abi_version(void);

/* ---- GEMM: every nn.Linear of the decoder and its backward (HF:81-83, 252-254, 279, 441).
 * C[M,N] = op(A) op(B) (+ R).  trans_a=0: A is [M][K]; 1: A is stored [K][M].  trans_b=0: B is [N][K]
 * (torch weight layout); 1: B is stored [K][N].  R (nullable, may alias C) is added in fp32. */
int sd_gemm_bf16(const void* A, const void* B, void* C, const void* R, int M, int N, int K, int64_t lda,
                 int64_t ldb, int64_t ldc, int64_t ldr, int trans_a, int trans_b, void* stream);

/* Split-K form for GEMMs with few output tiles and a very long K (lm_head dX: K = vocab): K slices
 * write fp32 slabs into `workspace`, summed in a fixed order by a second kernel (deterministic).
 * sd_gemm_splitk_plan returns the number of slices the library would use (1 = no split). */
int sd_gemm_splitk_plan(int M, int N, int K);
int64_t sd_gemm_splitk_workspace_bytes(int M, int N, int K);
int sd_gemm_bf16_splitk(const void* A, const void* B, void* C, const void* R, int M, int N, int K, int64_t lda,
                        int64_t ldb, int64_t ldc, int64_t ldr, int trans_a, int trans_b, void* workspace,
                        int64_t workspace_bytes, void* stream);

/* same, but the slabs are left un-reduced for a consumer that sums them (sd_rmsnorm_bwd_slabs); *nsplit_out = number
 * of slabs written to workspace ([nsplit][M][N] fp32), or 1 when no split was planned and bf16 C was written. */
int sd_gemm_bf16_splitk_partial(const void* A, const void* B, void* C, int M, int N, int K, int64_t lda, int64_t ldb,
                                int64_t ldc, int trans_a, int trans_b, void* workspace, int64_t workspace_bytes,
                                int* nsplit_out, void* stream);

/* o-projection dX (autograd of HF:279) with delta = rowsum(dO * O) per (token, head) -- the row constant of the
 * flash-attention backward (flash_attn, train.py:160,177) -- computed in its epilogue: d_ao [M,Hq*128] = dy [M,H] . Wo,
 * delta [B,Hq,T] fp32.  Pass the result to sd_attn_bwd2 with o = NULL ("delta is already filled").
 * SD_ERR_UNSUPPORTED: use sd_gemm_bf16 + sd_attn_bwd2 with o. */
int sd_gemm_odx_delta(const void* dy, const void* wo, void* d_ao, const void* o, int64_t ldo, float* delta, int M, int T,
                      int Hq, int H, void* stream);

/* Fused forward GEMMs (fall back to the separate launchers when they return SD_ERR_UNSUPPORTED):
 * sd_gemm_swiglu: act [M,I] = silu(x Wg^T) * (x Wu^T), wgu = [gate rows | up rows] [2I,K]; gu_out [M,2I] nullable (HF:81-83).
 * sd_gemm_qkv_rope: raw q|k|v [M,(Hq+2Hkv)*128] plus RMS-normalised + RoPE-rotated q|k [M,(Hq+Hkv)*128] (HF:252-257). */
int sd_gemm_swiglu(const void* x, const void* wgu, void* gu_out, void* act_out, int M, int I, int K, void* stream);
/* The four weight-gradient GEMMs of a decoder layer (autograd of HF:252-254, 279, 81-83) as ONE persistent launch:
 * C_p [M_p,N_p] = A_p^T . B_p, p < n <= 4, A_p = dY_p [K,M_p] (row stride lda), B_p = X_p [K,N_p], common K = tokens.
 * accumulate: 0 overwrite C_p, 1 C_p += (gradient accumulation).  Same results as n calls of
 * sd_gemm_bf16(trans_a=1, trans_b=1[, R = C]). */
typedef struct {
  const void* A;
  const void* B;
  void* C;
  int64_t lda, ldb, ldc;
  int32_t M, N;
} sd_gemm_problem;
int sd_gemm_grouped_tn(const sd_gemm_problem* probs, int n, int K, int accumulate, void* stream);
/* backward twin: d(gate|up) [M,2I] = SwiGLU'(gate_up) applied to d(act) = dy [M,H] . W_down [H,I], in the epilogue of
 * that GEMM (d(act) is never stored); equals sd_gemm_bf16 (NN) + sd_swiglu_bwd bit for bit. */
int sd_gemm_swiglu_bwd(const void* dy, const void* wdown, const void* gate_up, void* dgate_up, int M, int I, int H,
                       void* stream);
int sd_gemm_qkv_rope(const void* x, const void* wqkv, void* qkv_out, void* qk_out, const void* q_gain, const void* k_gain,
                     const void* cos_tab, const void* sin_tab, int M, int T, int Hq, int Hkv, int K, float eps,
                     void* stream);

/* ---- RMSNorm folded into the projection behind it (the FROZEN teacher, train.py:60-69 / 165-169: its weights never
 * change, so HF:59-64's  y = g * (x * rstd)  followed by  y W^T  (HF:252-254, 81-83) is  rstd * (x (W diag g)^T):
 * the gain is multiplied into the weight rows once at load and the row statistic is applied as a row scale in the
 * consuming GEMM's epilogue.  The statistic travels as per-128-column-tile partial sums of squares, fp32 [M, H/128]
 * (H % 512 == 0, H/128 <= 16), written by whoever produced the row and summed in a fixed order by the consumer:
 *   sd_embedding_fwd_ssq  the lookup of sd_embedding_fwd + the partials of its rows (input of layer 0);
 *   sd_gemm_bf16_ssq      C = A . B^T + R (the o / down projection with its residual, HF:304-323) + the partials of C;
 *   sd_gemm_qkv_rope_rs   sd_gemm_qkv_rope on the UN-normalised x with wqkv = Wqkv diag(input_layernorm gain);
 *   sd_gemm_swiglu_rs     sd_gemm_swiglu on the UN-normalised x with wgu = Wgu diag(post_attention_layernorm gain).
 * Arithmetic differs from the unfolded path only in rounding: the gain meets the weight instead of the activation, and
 * the normalised row is never rounded to bf16 (tolerance test: test_folded_teacher_forward_matches_unfolded). */
int sd_embedding_fwd_ssq(const int64_t* ids, const void* E, void* x, float* ssq_out, int M, int H, int V, void* stream);
int sd_gemm_bf16_ssq(const void* A, const void* B, void* C, const void* R, float* ssq_out, int M, int N, int K, int64_t lda,
                     int64_t ldb, int64_t ldc, int64_t ldr, void* stream);
int sd_gemm_qkv_rope_rs(const void* x, const void* wqkv, void* qkv_out, void* qk_out, const void* q_gain, const void* k_gain,
                        const void* cos_tab, const void* sin_tab, const float* ssq, int M, int T, int Hq, int Hkv, int K,
                        float eps, void* stream);
int sd_gemm_swiglu_rs(const void* x, const void* wgu, void* gu_out, void* act_out, const float* ssq, float eps, int M, int I,
                      int K, void* stream);

/* ---- RMSNorm (HF:59-64).  rstd (fp32 [M], nullable in fwd) is saved for backward. */
int sd_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int M, int H, float eps, void* stream);
int64_t sd_rmsnorm_bwd_workspace_bytes(int M, int H);
/* dx = d(norm)/dx . dy (+ dres, nullable: the residual-stream gradient); dw (+)= sum_rows dy * xhat */
int sd_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx,
                   void* dw, int accumulate_dw, void* workspace, int M, int H, void* stream);

/* *_bwd2: same, with the (tiny) gain-gradient reduce launched on `reduce_stream` after `event` (a hipEvent_t the
 * caller owns); both nullable = everything on `stream`.  The caller joins the streams before reusing `workspace`. */
int sd_rmsnorm_bwd2(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx,
                    void* dw, int accumulate_dw, void* workspace, int M, int H, void* reduce_stream, void* event,
                    void* stream);
/* dy given as nsplit fp32 slabs [nsplit][M][H] (un-reduced split-K output), summed in slab order inside the kernel */
int sd_rmsnorm_bwd_slabs(const float* dy_slabs, int nsplit, const void* x, const void* w, const float* rstd,
                         const void* dres, void* dx, void* dw, int accumulate_dw, void* workspace, int M, int H,
                         void* reduce_stream, void* event, void* stream);
int sd_qknorm_rope_bwd2(const void* dqk, const void* qkv, const void* q_gain, const void* k_gain, const void* cos_tab,
                        const void* sin_tab, void* dqkv, void* dq_gain, void* dk_gain, int accumulate_dw, void* workspace,
                        int M, int T, int Hq, int Hkv, float eps, void* reduce_stream, void* event, void* stream);

/* Deferred gain-gradient reduce: with dw == NULL (sd_rmsnorm_bwd2 / _slabs) or dq_gain == NULL (sd_qknorm_rope_bwd2)
 * the per-workgroup partial sums stay in `workspace` -- [sd_rmsnorm_bwd_partial_rows(M,H)][H] fp32, or
 * [sd_qknorm_rope_bwd_partial_rows(M,Hq,Hkv)][256] fp32 with the q gain in columns 0..127 and the k gain in
 * 128..255 -- and up to 8 such column sums are finished by ONE sd_colsum_reduce_batch launch (a layer's four gain
 * gradients: HF:59-64 weight of input_layernorm / post_attention_layernorm, HF:237-238 q_norm / k_norm).
 * out[c] (bf16) = (accumulate ? out[c] : 0) + sum_r partials[r*stride + c], fixed summation order. */
typedef struct {
  const float* partials;
  void* out;
  int nb, H, stride, accumulate;
} sd_colsum_problem;
int sd_rmsnorm_bwd_partial_rows(int M, int H);
int sd_qknorm_rope_bwd_partial_rows(int M, int Hq, int Hkv);
int sd_colsum_reduce_batch(const sd_colsum_problem* problems_host, int n, void* stream);

/* ---- per-head q/k RMSNorm (head_dim 128) then rotate-half RoPE (HF:252-257, 121-170).
 * qkv [M,(Hq+2Hkv)*128] -> qk_out [M,(Hq+Hkv)*128]; cos/sin tables bf16 [T,128]; token m has position m % T. */
int sd_qknorm_rope_fwd(const void* qkv, const void* q_gain, const void* k_gain, const void* cos_tab,
                       const void* sin_tab, void* qk_out, int M, int T, int Hq, int Hkv, float eps, void* stream);
int64_t sd_qknorm_rope_bwd_workspace_bytes(int M, int Hq, int Hkv);
int sd_qknorm_rope_bwd(const void* dqk, const void* qkv, const void* q_gain, const void* k_gain, const void* cos_tab,
                       const void* sin_tab, void* dqkv, void* dq_gain, void* dk_gain, int accumulate_dw, void* workspace,
                       int M, int T, int Hq, int Hkv, float eps, void* stream);

/* ---- SwiGLU (HF:81-83): gate_up [M,2I] (gate | up) -> act [M,I] = silu(gate)*up */
int sd_swiglu_fwd(const void* gate_up, void* act, int M, int I, void* stream);
int sd_swiglu_bwd(const void* dact, const void* gate_up, void* dgate_up, int M, int I, void* stream);

/* ---- embedding (HF:381) and its deterministic scatter-add backward (dE += scale * rows of dx) */
int sd_embedding_fwd(const int64_t* ids, const void* E, void* x, int M, int H, int V, void* stream);
int sd_embedding_bwd(const int64_t* ids, const void* dx, void* dE, int M, int H, int V, float scale, void* stream);
/* dst [M,H] = 0, then dst[rows[i]] = src[i] (rows unique): un-compacts the gradient of the rows the head kept
 * (inverse of the row gather sd_embedding_fwd(rows, table=x) ; distillation_loss.py:45 `x[valid]` run backwards) */
int sd_rows_scatter(const void* src, const int64_t* rows, void* dst, int n, int M, int H, void* stream);

/* ---- causal GQA flash attention, head_dim 128 (flash_attn via train.py:160,177; maths HF:185-207).
 * q [B*T, ldq] head hq at column hq*128; k, v likewise per kv head; o [B*T, ldo]; lse fp32 [B,Hq,T].
 * kv_len (int32 [B], nullable): keys >= kv_len[b] are masked (right padding). */
int sd_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int32_t* kv_len, int64_t ldq,
                int64_t ldk, int64_t ldv, int64_t ldo, int B, int T, int Hq, int Hkv, int head_dim, float scale,
                void* stream);
/* delta: fp32 [B,Hq,T] scratch (rowsum(dO*O)).  sd_attn_bwd2: same, with the dQ kernel launched on `side_stream`
 * (nullable) beside the dK/dV kernel -- they only share read-only inputs; `stream` has joined when it returns. */
int sd_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                float* delta, void* dq, void* dk, void* dv, const int32_t* kv_len, int64_t ldq, int64_t ldk, int64_t ldv,
                int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv, int B, int T, int Hq, int Hkv, int head_dim,
                float scale, void* stream);

int sd_attn_bwd2(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                 float* delta, void* dq, void* dk, void* dv, const int32_t* kv_len, int64_t ldq, int64_t ldk, int64_t ldv,
                 int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv, int B, int T, int Hq, int Hkv, int head_dim,
                 float scale, void* side_stream, void* stream);

/* ---- teacher log-softmax + top-K (train.py:80-91; extract_teacher_logits.py:114-129).
 * logits [rows, row_stride], first V columns used -> top_v fp16 [rows,K], top_i int32 [rows,K] sorted
 * descending, ties to the lowest index; lse_out fp32 [rows] nullable. */
int sd_logsoftmax_topk(const void* logits, void* top_v, void* top_i, float* lse_out, int rows, int64_t row_stride,
                       int V, int K, int dtype, void* stream);

/* ---- which rows the loss reads (distillation_loss.py:31-45: shift by one, labels != -100, optional speech mask).
 * labels int64 [B,T]; speech_mask int64 [B,T] nullable; mask_a / mask_b: attention masks int64 [B,T] (nullable) to
 * validate as valid-prefix (right-padded) masks in the same launch.  rows / row_labels: int64 [B*T] out, the first
 * meta[0] entries are the flat indices b*T+t in increasing order and the label each row predicts; meta[1] != 0 when a
 * mask has a 1 after a 0.  meta int32 [2] in device memory (the host reads it once: the row count sizes the lm_heads). */
int sd_loss_rows(const int64_t* labels, const int64_t* speech_mask, const int64_t* mask_a, const int64_t* mask_b,
                 int64_t* rows, int64_t* row_labels, int32_t* meta, int B, int T, void* stream);

/* ---- DistillationLoss.forward / backward (distillation_loss.py:14-128).
 * student_logits [B,T,V] (bf16 or fp32 per `dtype`); exactly one of teacher_logits [B,T,V] (dense) or
 * (top_k_v fp16, top_k_i int32) [B,T,K] (sparse); labels int64 [B,T]; speech_mask uint8 [B,T] nullable.
 * loss_out fp32[8] = {total, task, distill, teacher, N_valid, n_hits, 0, 0}; row_stats: scratch of
 * sd_kdloss_stats_bytes(B,T) kept from fwd to bwd.  grad_total: fp32[1] upstream gradient (nullable = 1).
 * grad_logits may alias student_logits. */
int64_t sd_kdloss_stats_bytes(int B, int T);
int sd_kdloss_fwd(const void* student_logits, const void* teacher_logits, const void* top_k_v, const void* top_k_i,
                  const int64_t* labels, const uint8_t* speech_mask, void* row_stats, float* loss_out, int B, int T, int V,
                  int K, float temperature, float alpha, int dtype, void* stream);
int sd_kdloss_bwd(const void* student_logits, const void* teacher_logits, const void* top_k_v, const void* top_k_i,
                  const int64_t* labels, const void* row_stats, const float* loss_out, const float* grad_total,
                  void* grad_logits, int B, int T, int V, int K, float temperature, float alpha, int dtype, void* stream);

/* Same loss on rows the caller already shifted and selected (distillation_loss.py:31-45 done by the caller):
 * student_logits [R,V], teacher_logits [R,V] or (top_k_v, top_k_i) [R,K], row_labels int64 [R] = the label each row
 * predicts (-100 still masks a row); row_stats of sd_kdloss_stats_bytes(R,1). */
int sd_kdloss_fwd_rows(const void* student_logits, const void* teacher_logits, const void* top_k_v, const void* top_k_i,
                       const int64_t* row_labels, void* row_stats, float* loss_out, int R, int V, int K,
                       float temperature, float alpha, int dtype, void* stream);
int sd_kdloss_bwd_rows(const void* student_logits, const void* teacher_logits, const void* top_k_v, const void* top_k_i,
                       const int64_t* row_labels, const void* row_stats, const float* loss_out, const float* grad_total,
                       void* grad_logits, int R, int V, int K, float temperature, float alpha, int dtype, void* stream);

/* ---- fused AdamW on bf16 params with bf16 state (HF Trainer default optimizer on the bf16 student,
 * train.py:174,331-354; quirk Q5) and the global grad-norm / clip (HF trainer max_grad_norm). */
/* out_accum[0] += sum x^2; `partials`: SD_SUMSQ_PARTIALS floats of caller-owned device scratch holding the per-workgroup
 * sums between the two launches (fixed-order reduction: the norm is bit-identical on every data-parallel rank; calls
 * that may overlap on different streams pass different scratch). */
#define SD_SUMSQ_PARTIALS 2048
int sd_sumsq_bf16(const void* x, int64_t n, float* out_accum, float* partials, void* stream);
int sd_adamw_bf16(void* param, const void* grad, void* exp_avg, void* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, const float* grad_sumsq, float max_grad_norm,
                  void* stream);

/* ---- LoRA student (train.py:180-202: peft LoRA, r = 32, on q/k/v/o/gate/up/down_proj of every decoder layer;
 * parity unpinned -- peft is an unvendored dependency, oracle/lora.py restates its published layer).
 * The adapter is applied to the MERGED weight: W_eff = W_res + (s B) A once per optimizer step (sd_lora_merge), the
 * decoder backward writes the full weight gradient dW, and dA = (s B)^T dW, dB = dW (s A)^T (sd_lora_project).  All
 * targets go in ONE launch each, driven by a plan the host builds once.
 *   A [r_pad, in] and B [out, r_pad] are fp32 masters (rows / columns r..r_pad are zero and stay zero; r_pad = r
 *   rounded up to 32, 64 or 128); the kernels read their bf16 shadows, written by sd_adamw_f32_shadow:
 *   a_shadow = bf16(A), a_scaled = bf16(s A), b_scaled = bf16(s B).  d_a / d_b: bf16, same shapes as A / B.
 * Shapes: in_features % 128 == 0, out_features % 32 == 0, else SD_ERR_UNSUPPORTED; every pointer 16-byte aligned. */
typedef struct {
  const void* w_res;   /* [out, in] bf16: frozen (residual) base weight */
  void* w_out;         /* [out, in] bf16: merged weight the decoder reads */
  const void* w_grad;  /* [out, in] bf16: gradient w.r.t. the merged weight */
  const void* a_shadow;
  const void* a_scaled;
  const void* b_scaled;
  void* d_a;
  void* d_b;
  int32_t out_features, in_features;
} SdLoraTarget;
int64_t sd_lora_plan_bytes(int n_targets);
/* Fills `plan_host` (host memory, sd_lora_plan_bytes(n) bytes); the caller copies it to the device once. */
int sd_lora_plan_build(const SdLoraTarget* targets, int n_targets, int r_pad, void* plan_host, int64_t plan_bytes);
int sd_lora_merge(const void* plan_dev, const void* plan_host, void* stream);
int sd_lora_project(const void* plan_dev, const void* plan_host, void* stream);
/* AdamW on fp32 parameters with fp32 moments and a bf16 gradient (same update rule and clip as sd_adamw_bf16), also
 * writing shadow = bf16(p) and shadow_scaled = bf16(scale * p).  n % 4 == 0. */
int sd_adamw_f32_shadow(float* param, const void* grad, float* exp_avg, float* exp_avg_sq, void* shadow,
                        void* shadow_scaled, float scale, int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, const float* grad_sumsq, float max_grad_norm, void* stream);

/* ---- whole-decoder runner: Qwen3ForCausalLM forward (HF:381-441) / backward, one C call each.
 * Weight layout: q|k|v projections fused row-wise into wqkv [(Hq+2Hkv)*128, h]; gate|up into wgu [2I, h]. */
typedef struct {
  int32_t vocab, hidden, inter, layers, n_q, n_kv, head_dim, tied;
  float eps;
  int32_t pad_;
} sd_qwen3_dims;

typedef struct {
  void *wqkv, *wo, *wgu, *wdown, *q_gain, *k_gain, *ln1, *ln2;
} sd_qwen3_layer;

typedef struct {
  void* embed;      /* [V,h] */
  void* lm_head;    /* [V,h]; == embed when tied */
  void* final_norm; /* [h] */
  const sd_qwen3_layer* layers_host; /* HOST array of `layers` entries holding DEVICE pointers */
} sd_qwen3_params;

/* What a forward keeps in `acts` (the `save_for_backward` argument below):
 *   SD_SAVE_NONE          inference (the frozen teacher): one layer's buffers + a ping-pong residual stream;
 *   SD_SAVE_ALL           every activation the backward reads, for all layers (default training mode);
 *   SD_SAVE_LAYER_INPUTS  layer-granular recompute = what `gradient_checkpointing_enable()` means in HF
 *                         (train.py:204-208, modeling_qwen3.py decoder-layer checkpointing): only the residual stream
 *                         entering each layer + two layer work sets; the backward (SD_BWD_RECOMPUTE) runs a layer's
 *                         forward again right before its backward.  Same kernels on the same inputs, so the
 *                         gradients are bit-identical to SD_SAVE_ALL.
 *   SD_SAVE_NONE_FOLDED   inference with every decoder layer's two RMSNorm gains folded into the weights the caller passes:
 *                         layers_host[l].wqkv = Wqkv diag(ln1), .wgu = Wgu diag(ln2) (bf16(W[n][k] * g[k]), done once at
 *                         load); ln1 / ln2 are not read and no RMSNorm kernel runs except the final one (its consumer is
 *                         the tied embedding table): 1 instead of 2 L + 1 norm launches per pass.  Only for dims with
 *                         sd_qwen3_fold_supported(d) != 0; logits agree with SD_SAVE_NONE to bf16 rounding. */
#define SD_SAVE_NONE 0
#define SD_SAVE_ALL 1
#define SD_SAVE_LAYER_INPUTS 2
#define SD_SAVE_NONE_FOLDED 3
/* OR-ed into `save` of sd_qwen3_forward(_rows): the caller runs ANOTHER pass beside this one on a second stream (the frozen
 * teacher beside the student's forward, train.py:60-69 vs :54).  Launches are then sized for CU-time per FLOP instead of
 * for covering every CU on their own (larger tiles on fewer workgroups for the N = hidden projections); results agree with
 * the unflagged pass to bf16 rounding.  Without a second stream the flag costs time: leave it off for a pass that runs alone. */
#define SD_FWD_CONCURRENT 0x100
int sd_qwen3_fold_supported(const sd_qwen3_dims* d);
/* bytes of activation storage for `sd_qwen3_forward` in that mode (negative: SD_ERR_* for an unknown mode) */
int64_t sd_qwen3_acts_bytes(const sd_qwen3_dims* d, int B, int T, int save_for_backward);
int64_t sd_qwen3_bwd_scratch_bytes(const sd_qwen3_dims* d, int B, int T);

/* ids int64 [B,T]; kv_len int32 [B] nullable; cos/sin bf16 [T,128]; acts: caller buffer of
 * sd_qwen3_acts_bytes; logits bf16 [B*T, V] out (nullable: stop after the final norm);
 * hidden_out bf16 [B*T,h] nullable */
int sd_qwen3_forward(const sd_qwen3_dims* d, const sd_qwen3_params* p, const int64_t* ids, const int32_t* kv_len,
                     const void* cos_tab, const void* sin_tab, void* acts, int64_t acts_bytes, void* logits, int B, int T,
                     int save_for_backward, void* stream);
/* same, with the lm_head applied only to the rows listed in head_rows (int64 [n_head_rows], flat b*T+t indices,
 * unique, device memory; NULL = all rows): logits is then bf16 [n_head_rows, V].  The training step needs only the
 * rows whose shifted label is not -100 (distillation_loss.py:37-45); HF computes all of them (train.py:54-55). */
int sd_qwen3_forward_rows(const sd_qwen3_dims* d, const sd_qwen3_params* p, const int64_t* ids, const int32_t* kv_len,
                          const void* cos_tab, const void* sin_tab, void* acts, int64_t acts_bytes, void* logits,
                          const int64_t* head_rows, int n_head_rows, int B, int T, int save_for_backward, void* stream);
/* g: same structure as p but holding gradient buffers (bf16, same shapes); dlogits bf16 [B*T,V];
 * accumulate: bit set of SD_BWD_ACCUMULATE (add to the gradient buffers: gradient accumulation; otherwise overwrite)
 * and SD_BWD_RECOMPUTE (`acts` was written by a forward with SD_SAVE_LAYER_INPUTS).
 * on_grads_ready (nullable) is called ON THE HOST, from inside this call, each time the kernels that
 * finish one group of gradients have been enqueued on `stream`: stage = SD_STAGE_HEAD (lm_head dW
 * + final norm, before any layer), layer index L-1..0 (that layer's 8 tensors), SD_STAGE_EMBED (the
 * embedding scatter-add, last).  A data-parallel caller records an event there and starts that
 * bucket's RCCL all-reduce on a second stream, overlapping it with the rest of backward.
 * side_stream (nullable hipStream_t): when given, the weight-gradient GEMMs of a layer (dW = dY^T X, which
 * nothing else in the layer depends on) are launched there and overlap the dX chain on `stream`; ordering
 * is by HIP events, `stream` has waited for all of them when the call returns (and before each callback).
 * dx0_out (nullable, bf16 [B*T,h]): when given, the gradient w.r.t. the embedding OUTPUT is written there and
 * the local embedding scatter-add is skipped -- the data-parallel caller then reduces the dense (lm_head)
 * part of the tied gradient early and exchanges only the B*T touched rows (ddp.py). */
#define SD_STAGE_HEAD (-1)
#define SD_STAGE_EMBED (-2)
#define SD_BWD_ACCUMULATE 1
#define SD_BWD_RECOMPUTE 2
typedef void (*sd_stage_cb)(int stage, void* user);
int sd_qwen3_backward(const sd_qwen3_dims* d, const sd_qwen3_params* p, const sd_qwen3_params* g, const int64_t* ids,
                      const int32_t* kv_len, const void* cos_tab, const void* sin_tab, void* acts, int64_t acts_bytes,
                      void* dlogits, void* scratch, int64_t scratch_bytes, int B, int T, int accumulate,
                      void* dx0_out, sd_stage_cb on_grads_ready, void* cb_user, void* side_stream, void* stream);
/* same for a forward made by sd_qwen3_forward_rows with the same head_rows: dlogits is bf16 [n_head_rows, V] */
int sd_qwen3_backward_rows(const sd_qwen3_dims* d, const sd_qwen3_params* p, const sd_qwen3_params* g, const int64_t* ids,
                           const int32_t* kv_len, const void* cos_tab, const void* sin_tab, void* acts,
                           int64_t acts_bytes, void* dlogits, const int64_t* head_rows, int n_head_rows, void* scratch,
                           int64_t scratch_bytes, int B, int T, int accumulate, void* dx0_out,
                           sd_stage_cb on_grads_ready, void* cb_user, void* side_stream, void* stream);

/* ---- stream placement.  HIP multiplexes streams onto a few hardware queues (4 by default); streams that share a
 * queue never overlap.  Measures, with a `spin_us`-long busy-wait kernel on stream_a and an empty one on stream_b,
 * whether work on b runs while a is busy: *overlap = 1/0.  Synchronises both streams (a calibration call, made once
 * when the host picks its side streams: teacher beside student (train.py:60-69 vs :54), dW beside dX, RCCL beside
 * backward). */
int sd_streams_overlap(void* stream_a, void* stream_b, float spin_us, int* overlap);

/* ---- optional live timing (bench.py): HIP events around every launch, on the launch stream.
 * kinds index the arrays of sd_prof_end; work = algorithmic FLOPs (GEMM, attention) or bytes (others). */
enum {
  SD_K_GEMM_NT = 0, SD_K_GEMM_NN, SD_K_GEMM_TN, SD_K_ATTN_FWD, SD_K_ATTN_BWD_DKV, SD_K_ATTN_BWD_DQ, SD_K_LOSS_FWD,
  SD_K_LOSS_BWD, SD_K_TOPK, SD_K_RMSNORM, SD_K_QKROPE, SD_K_SWIGLU, SD_K_EMBED, SD_K_OPTIM, SD_K_MISC, SD_K_GEMM_NT_STAG,
  SD_K_COUNT
};
int sd_prof_begin(void);
int sd_prof_end(double* ms, double* work, int64_t* count, int n_kinds);
/* The same measurement per KERNEL SYMBOL (as rocprofv3 --kernel-trace names it, without namespace and arguments), as
 * text lines "symbol<TAB>kind<TAB>ms<TAB>work<TAB>launches"; returns the bytes needed (cap = 0 sizes the buffer). */
int64_t sd_prof_symbols(char* buf, int64_t cap);

#ifdef __cplusplus
}
#endif
