"""Tensor-level wrappers over the C ABI (one function per launcher in include/sd_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every function takes CUDA(HIP)
tensors, passes raw pointers + sizes + torch's current stream to libsd_hip.so and converts a
non-zero status to an exception.  CPU tensors are rejected -- there is no fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from ._lib import check, load_lib

BF16, F32 = 0, 1


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


# ---------------------------------------------------------------------------------------------- side streams
# HIP maps streams onto a few hardware queues (4 by default); two streams on one queue run strictly in turn.  Which
# streams alias depends on how many the process created before (torch pre-creates pools of 32; torch.distributed and
# RCCL add their own), so the side streams of the step are chosen by measurement: sd_streams_overlap runs a short
# busy-wait on one stream and checks that the other is not held up behind it.
_SIDE_STREAMS = {}  # (device index, role) -> torch.cuda.Stream
_ROLES_ACTIVE_TOGETHER = {"teacher": (), "dw": ("comm",), "comm": ("dw",), "h2d": ("teacher",)}  # besides the main stream


def streams_overlap(a, b, spin_us=200.0):
    """True when work on stream ``b`` executes while stream ``a`` is busy (they sit on different hardware queues)."""
    out = C.c_int(0)
    check(load_lib().sd_streams_overlap(a.cuda_stream, b.cuda_stream, spin_us, C.byref(out)), "sd_streams_overlap")
    return bool(out.value)


def concurrent_stream(device, role):
    """The process-wide side stream of ``role`` on ``device``: "teacher" (frozen teacher beside the student forward),
    "dw" (weight-gradient GEMMs beside the dX chain), "comm" (gradient all-reduce beside backward), "h2d" (the next
    accumulation window's batches, copied in while the optimizer step is still running: DistillationTrainer.get_batch_samples).  Picked once, by
    experiment, so that it overlaps the current (main) stream and the roles that are busy at the same time; the
    environment variable SD_STREAM_PICK=0 takes the first stream torch hands out instead."""
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), role)
    if key in _SIDE_STREAMS:
        return _SIDE_STREAMS[key]
    with torch.cuda.device(device):
        main = torch.cuda.current_stream()
        prio = int(os.environ.get("SD_STREAM_PRIORITY_" + role.upper(), "0"))  # measurement: -1 = high

        def new_stream():
            return torch.cuda.Stream(priority=prio)
        chosen = new_stream()
        if os.environ.get("SD_STREAM_PICK", "1") != "0":
            peers = [_SIDE_STREAMS[(key[0], r)] for r in _ROLES_ACTIVE_TOGETHER[role] if (key[0], r) in _SIDE_STREAMS]
            fallback = None
            for _ in range(32):  # torch hands out its pool of 32 streams round-robin
                if streams_overlap(main, chosen):
                    if all(streams_overlap(p, chosen) for p in peers):
                        break
                    fallback = fallback or chosen
                chosen = new_stream()
            else:
                chosen = fallback or chosen  # no stream clear of every peer: at least clear of the main stream
    _SIDE_STREAMS[key] = chosen
    return chosen


def _need(t, dtype=None, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"speech_distill_amd: {name} must be a GPU tensor (no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def _dt(t):
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f"unsupported dtype {t.dtype} (bf16 or fp32)")


# ------------------------------------------------------------------------------------------ GEMM
def gemm(a, b, trans_a=False, trans_b=False, residual=None, out=None, split_k=False):
    """C = op(a) @ op(b) (+ residual).  trans_a: a is stored [K,M]; trans_b=False: b is [N,K]."""
    _need(a, torch.bfloat16, "a"), _need(b, torch.bfloat16, "b")
    K, M = (a.shape if trans_a else a.shape[::-1])
    if trans_b:
        Kb, N = b.shape
    else:
        N, Kb = b.shape
    if Kb != K:
        raise ValueError(f"gemm: contraction mismatch {K} vs {Kb}")
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    r = None if residual is None else _need(residual, torch.bfloat16, "residual")
    if split_k:
        lib = load_lib()
        nb = lib.sd_gemm_splitk_workspace_bytes(M, N, K)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=a.device)
        check(lib.sd_gemm_bf16_splitk(a.data_ptr(), b.data_ptr(), out.data_ptr(), _p(r), M, N, K, a.stride(0), b.stride(0),
                                      out.stride(0), 0 if r is None else r.stride(0), int(trans_a), int(trans_b),
                                      ws.data_ptr(), nb, _stream()), "sd_gemm_bf16_splitk")
        return out
    check(load_lib().sd_gemm_bf16(a.data_ptr(), b.data_ptr(), out.data_ptr(), _p(r), M, N, K, a.stride(0), b.stride(0),
                                  out.stride(0), 0 if r is None else r.stride(0), int(trans_a), int(trans_b), _stream()),
          "sd_gemm_bf16")
    return out


# ------------------------------------------------------------------------------------- elementwise
def rmsnorm_fwd(x, w, eps=1e-6):
    _need(x, torch.bfloat16, "x"), _need(w, torch.bfloat16, "w")
    M, H = x.shape
    y = torch.empty_like(x)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    check(load_lib().sd_rmsnorm_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), rstd.data_ptr(), M, H, eps, _stream()),
          "sd_rmsnorm_fwd")
    return y, rstd


def rmsnorm_bwd(dy, x, w, rstd, dres=None, dw=None, accumulate=False):
    M, H = x.shape
    lib = load_lib()
    ws = torch.empty(lib.sd_rmsnorm_bwd_workspace_bytes(M, H), dtype=torch.uint8, device=x.device)
    dx = torch.empty_like(x)
    if dw is None:
        dw = torch.zeros_like(w)
    check(lib.sd_rmsnorm_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), rstd.data_ptr(), _p(dres), dx.data_ptr(),
                             dw.data_ptr(), int(accumulate), ws.data_ptr(), M, H, _stream()), "sd_rmsnorm_bwd")
    return dx, dw


def rope_tables(T, device, theta=1e6, d=128):
    """cos/sin [T,d] exactly as HF computes them (modeling_qwen3.py:113-137), rounded to bf16."""
    inv = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    fr = torch.arange(T, dtype=torch.float32)[:, None] * inv[None, :]
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos().to(torch.bfloat16).to(device), emb.sin().to(torch.bfloat16).to(device)


def qknorm_rope_fwd(qkv, q_gain, k_gain, cos, sin, T, Hq, Hkv, eps=1e-6):
    M = qkv.shape[0]
    out = torch.empty(M, (Hq + Hkv) * 128, dtype=torch.bfloat16, device=qkv.device)
    check(load_lib().sd_qknorm_rope_fwd(qkv.data_ptr(), q_gain.data_ptr(), k_gain.data_ptr(), cos.data_ptr(),
                                        sin.data_ptr(), out.data_ptr(), M, T, Hq, Hkv, eps, _stream()),
          "sd_qknorm_rope_fwd")
    return out


def qknorm_rope_bwd(dqk, qkv, q_gain, k_gain, cos, sin, T, Hq, Hkv, eps=1e-6):
    M = qkv.shape[0]
    lib = load_lib()
    ws = torch.empty(lib.sd_qknorm_rope_bwd_workspace_bytes(M, Hq, Hkv), dtype=torch.uint8, device=qkv.device)
    dqkv = torch.zeros_like(qkv)
    dqg, dkg = torch.zeros_like(q_gain), torch.zeros_like(k_gain)
    check(lib.sd_qknorm_rope_bwd(dqk.data_ptr(), qkv.data_ptr(), q_gain.data_ptr(), k_gain.data_ptr(), cos.data_ptr(),
                                 sin.data_ptr(), dqkv.data_ptr(), dqg.data_ptr(), dkg.data_ptr(), 0, ws.data_ptr(), M, T,
                                 Hq, Hkv, eps, _stream()), "sd_qknorm_rope_bwd")
    return dqkv, dqg, dkg


def swiglu_fwd(gu):
    M, I2 = gu.shape
    act = torch.empty(M, I2 // 2, dtype=torch.bfloat16, device=gu.device)
    check(load_lib().sd_swiglu_fwd(gu.data_ptr(), act.data_ptr(), M, I2 // 2, _stream()), "sd_swiglu_fwd")
    return act


def swiglu_bwd(dact, gu):
    M, I2 = gu.shape
    dgu = torch.empty_like(gu)
    check(load_lib().sd_swiglu_bwd(dact.data_ptr(), gu.data_ptr(), dgu.data_ptr(), M, I2 // 2, _stream()), "sd_swiglu_bwd")
    return dgu


def embedding_fwd(ids, E):
    M = ids.numel()
    V, H = E.shape
    x = torch.empty(M, H, dtype=torch.bfloat16, device=E.device)
    check(load_lib().sd_embedding_fwd(ids.data_ptr(), E.data_ptr(), x.data_ptr(), M, H, V, _stream()), "sd_embedding_fwd")
    return x


def embedding_bwd(ids, dx, dE, scale=1.0):
    V, H = dE.shape
    check(load_lib().sd_embedding_bwd(ids.data_ptr(), dx.data_ptr(), dE.data_ptr(), ids.numel(), H, V, float(scale),
                                      _stream()),
          "sd_embedding_bwd")
    return dE


# --------------------------------------------------------------------------------------- attention
def attn_fwd(q, k, v, B, T, Hq, Hkv, kv_len=None):
    """q [B*T, Hq*128], k/v [B*T, Hkv*128] (any row stride, unit column stride)."""
    M = B * T
    o = torch.empty(M, Hq * 128, dtype=torch.bfloat16, device=q.device)
    lse = torch.empty(B, Hq, T, dtype=torch.float32, device=q.device)
    check(load_lib().sd_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), _p(kv_len),
                                 q.stride(0), k.stride(0), v.stride(0), o.stride(0), B, T, Hq, Hkv, 128, 128 ** -0.5,
                                 _stream()), "sd_attn_fwd")
    return o, lse


def attn_bwd(q, k, v, o, do, lse, B, T, Hq, Hkv, kv_len=None, delta=None):
    """o = None: `delta` already holds rowsum(dO * O) per (batch, head, token) (gemm_odx_delta)."""
    if o is None and delta is None:
        raise ValueError("attn_bwd: either o or a precomputed delta")
    if delta is None:
        delta = torch.empty_like(lse)
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    check(load_lib().sd_attn_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), _p(o), do.data_ptr(), lse.data_ptr(),
                                 delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), _p(kv_len), q.stride(0),
                                 k.stride(0), v.stride(0), do.stride(0), dq.stride(0), dk.stride(0), dv.stride(0), B, T, Hq,
                                 Hkv, 128, 128 ** -0.5, _stream()), "sd_attn_bwd")
    return dq, dk, dv


# ------------------------------------------------------------------------------------ top-K / loss
def logsoftmax_topk(logits, k, vocab_size=None):
    """train.py:80-91: logits[..., :vocab] -> log_softmax -> topk(k) -> (fp16 values, int32 indices)."""
    _need(logits, None, "logits")
    lead = logits.shape[:-1]
    Vt = logits.shape[-1]
    V = Vt if vocab_size is None else min(int(vocab_size), Vt)
    if (Vt % 8) or (logits.data_ptr() % 16):
        # rows must start 16-byte aligned for the vector loads: take the truncated copy the
        # reference itself makes at train.py:82-83
        logits = logits[..., :V].contiguous()
        Vt = V
    rows = logits.numel() // Vt
    tv = torch.empty(*lead, k, dtype=torch.float16, device=logits.device)
    ti = torch.empty(*lead, k, dtype=torch.int32, device=logits.device)
    check(load_lib().sd_logsoftmax_topk(logits.data_ptr(), tv.data_ptr(), ti.data_ptr(), 0, rows, Vt, V, k, _dt(logits),
                                        _stream()), "sd_logsoftmax_topk")
    return tv, ti


class KDLossFn(torch.autograd.Function):
    """DistillationLoss.forward + its backward on the HIP kernels (distillation_loss.py:14-128)."""

    @staticmethod
    def forward(ctx, student_logits, labels, teacher_logits, top_v, top_i, speech_mask, temperature, alpha, inplace_grad):
        s = _need(student_logits, None, "student_logits")
        B, T, V = s.shape
        dt = _dt(s)
        lib = load_lib()
        labels = _need(labels.to(torch.int64), torch.int64, "labels")
        K = 0
        if teacher_logits is not None:
            teacher_logits = _need(teacher_logits.to(s.dtype), None, "teacher_logits")
            if teacher_logits.shape != s.shape:
                raise ValueError(f"teacher_logits {tuple(teacher_logits.shape)} != student_logits {tuple(s.shape)}")
            top_v = top_i = None
        elif top_v is not None and top_i is not None:
            K = top_v.shape[-1]
            # distillation_loss.py:78-91 moves the pre-extracted arrays to the student's device
            top_v = top_v.to(device=s.device, dtype=torch.float16).contiguous()
            top_i = top_i.to(device=s.device, dtype=torch.int32).contiguous()
        else:
            raise ValueError("Either teacher_logits or top_k must be provided")  # distillation_loss.py:120
        mask = None
        if speech_mask is not None:
            mask = (speech_mask.to(s.device) != 0).to(torch.uint8).contiguous()
        stats = torch.empty(lib.sd_kdloss_stats_bytes(B, T), dtype=torch.uint8, device=s.device)
        out = torch.empty(8, dtype=torch.float32, device=s.device)
        check(lib.sd_kdloss_fwd(s.data_ptr(), _p(teacher_logits), _p(top_v), _p(top_i), labels.data_ptr(), _p(mask),
                                stats.data_ptr(), out.data_ptr(), B, T, V, K, float(temperature), float(alpha), dt,
                                _stream()), "sd_kdloss_fwd")
        ctx.save_for_backward(s, labels, teacher_logits, top_v, top_i, stats, out)
        ctx.cfg = (B, T, V, K, float(temperature), float(alpha), dt, bool(inplace_grad))
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g_total, _g_out):
        s, labels, teacher_logits, top_v, top_i, stats, out = ctx.saved_tensors
        B, T, V, K, temperature, alpha, dt, inplace = ctx.cfg
        grad = s if inplace else torch.empty_like(s)
        go = g_total.to(torch.float32).reshape(1).contiguous()
        check(load_lib().sd_kdloss_bwd(s.data_ptr(), _p(teacher_logits), _p(top_v), _p(top_i), labels.data_ptr(),
                                       stats.data_ptr(), out.data_ptr(), go.data_ptr(), grad.data_ptr(), B, T, V, K,
                                       temperature, alpha, dt, _stream()), "sd_kdloss_bwd")
        return grad, None, None, None, None, None, None, None, None


def gemm_grouped_tn(pairs, accumulate_into=None):
    """[(dY [K,M], X [K,N]), ...] (at most 4, common K) -> [dY^T @ X [M,N], ...] in one persistent launch;
    accumulate_into: list of [M,N] tensors to add to in place."""
    from ._lib import GemmProblem
    n = len(pairs)
    probs = (GemmProblem * n)()
    outs = []
    K = pairs[0][0].shape[0]
    for i, (a, b) in enumerate(pairs):
        _need(a, torch.bfloat16, "dY"), _need(b, torch.bfloat16, "X")
        if a.shape[0] != K or b.shape[0] != K:
            raise ValueError("gemm_grouped_tn: every problem must have the same contraction length")
        c = (torch.empty(a.shape[1], b.shape[1], dtype=torch.bfloat16, device=a.device) if accumulate_into is None
             else accumulate_into[i])
        outs.append(c)
        probs[i] = GemmProblem(a.data_ptr(), b.data_ptr(), c.data_ptr(), a.stride(0), b.stride(0), c.stride(0), a.shape[1],
                               b.shape[1])
    check(load_lib().sd_gemm_grouped_tn(C.cast(probs, C.c_void_p), n, K, int(accumulate_into is not None), _stream()),
          "sd_gemm_grouped_tn")
    return outs


def gemm_swiglu_bwd(dy, wdown, gate_up):
    """d(gate|up) [M,2I] from dy [M,h], W_down [h,I] (torch layout [out=h, in=I]) and the forward's gate|up."""
    _need(dy, torch.bfloat16, "dy")
    M, H = dy.shape
    I = wdown.shape[1]
    out = torch.empty(M, 2 * I, dtype=torch.bfloat16, device=dy.device)
    check(load_lib().sd_gemm_swiglu_bwd(dy.data_ptr(), wdown.data_ptr(), gate_up.data_ptr(), out.data_ptr(), M, I, H,
                                        _stream()), "sd_gemm_swiglu_bwd")
    return out


def gemm_odx_delta(dy, wo, o, T, Hq):
    """d(attention output) [M, Hq*128] = dy [M,H] @ wo [H, Hq*128] and delta [M/T, Hq, T] = rowsum(d_ao * o) per head,
    the second in the GEMM's epilogue (sd_gemm_odx_delta)."""
    _need(dy, torch.bfloat16, "dy"), _need(wo, torch.bfloat16, "wo"), _need(o, torch.bfloat16, "o")
    M, H = dy.shape
    dao = torch.empty(M, Hq * 128, dtype=torch.bfloat16, device=dy.device)
    delta = torch.empty(M // T, Hq, T, dtype=torch.float32, device=dy.device)
    check(load_lib().sd_gemm_odx_delta(dy.data_ptr(), wo.data_ptr(), dao.data_ptr(), o.data_ptr(), o.stride(0),
                                       delta.data_ptr(), M, T, Hq, H, _stream()), "sd_gemm_odx_delta")
    return dao, delta


class KDLossRowsFn(torch.autograd.Function):
    """The same loss on rows the caller has already shifted and selected (distillation_loss.py:31-45 done by the
    caller): student_logits [R,V], row_labels [R], teacher_logits [R,V] or (top_v, top_i) [R,K]."""

    @staticmethod
    def forward(ctx, student_logits, row_labels, teacher_logits, top_v, top_i, temperature, alpha, inplace_grad):
        s = _need(student_logits, None, "student_logits")
        R, V = s.shape
        dt = _dt(s)
        lib = load_lib()
        row_labels = _need(row_labels.to(torch.int64), torch.int64, "row_labels")
        K = 0
        if teacher_logits is not None:
            teacher_logits = _need(teacher_logits.to(s.dtype), None, "teacher_logits")
            if teacher_logits.shape != s.shape:
                raise ValueError(f"teacher_logits {tuple(teacher_logits.shape)} != student_logits {tuple(s.shape)}")
            top_v = top_i = None
        elif top_v is not None and top_i is not None:
            K = top_v.shape[-1]
            top_v = top_v.to(device=s.device, dtype=torch.float16).contiguous()
            top_i = top_i.to(device=s.device, dtype=torch.int32).contiguous()
        else:
            raise ValueError("Either teacher_logits or top_k must be provided")  # distillation_loss.py:120
        stats = torch.empty(lib.sd_kdloss_stats_bytes(R, 1), dtype=torch.uint8, device=s.device)
        out = torch.empty(8, dtype=torch.float32, device=s.device)
        check(lib.sd_kdloss_fwd_rows(s.data_ptr(), _p(teacher_logits), _p(top_v), _p(top_i), row_labels.data_ptr(),
                                     stats.data_ptr(), out.data_ptr(), R, V, K, float(temperature), float(alpha), dt,
                                     _stream()), "sd_kdloss_fwd_rows")
        ctx.save_for_backward(s, row_labels, teacher_logits, top_v, top_i, stats, out)
        ctx.cfg = (R, V, K, float(temperature), float(alpha), dt, bool(inplace_grad))
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g_total, _g_out):
        s, row_labels, teacher_logits, top_v, top_i, stats, out = ctx.saved_tensors
        R, V, K, temperature, alpha, dt, inplace = ctx.cfg
        grad = s if inplace else torch.empty_like(s)
        go = g_total.to(torch.float32).reshape(1).contiguous()
        check(load_lib().sd_kdloss_bwd_rows(s.data_ptr(), _p(teacher_logits), _p(top_v), _p(top_i), row_labels.data_ptr(),
                                            stats.data_ptr(), out.data_ptr(), go.data_ptr(), grad.data_ptr(), R, V, K,
                                            temperature, alpha, dt, _stream()), "sd_kdloss_bwd_rows")
        return grad, None, None, None, None, None, None, None


def left_padded(attention_mask):
    """0-d bool tensor: some row has a 1 after a 0, i.e. the mask is not a valid-prefix (right-padded) mask."""
    am = attention_mask != 0
    return (am[:, 1:] & ~am[:, :-1]).any()


def loss_rows(labels, speech_mask=None, right_padded=()):
    """Flat indices b*T+t of the rows the loss reads, and the label each one predicts: position t < T-1 whose
    NEXT label is not -100 (and whose next mask bit is set) -- distillation_loss.py:31-45.  One host read (the
    row count sizes the lm_head GEMMs); ``right_padded``: attention masks to validate in that same read -- raises
    ValueError if one of them is not a valid-prefix mask (see HipQwen3ForCausalLM.forward)."""
    B, T = labels.shape
    masks = [m for m in right_padded if m is not None]
    # every mask is indexed as [B, T] with the LABELS' B and T (the collator pads teacher and student separately,
    # data.py:219-278: a teacher mask of another length is not this batch's grid)
    for what, m in [("attention mask", m) for m in masks] + ([("speech_token_mask", speech_mask)] if speech_mask is not None else []):
        if tuple(m.shape) != (B, T):
            raise ValueError(f"loss_rows: {what} {tuple(m.shape)} != labels {(B, T)}")
    if labels.is_cuda and len(masks) <= 2:  # one launch (sd_loss_rows) + one 8-byte read
        def i64(t):  # masks: any dtype, non-zero = 1
            t = t.to(labels.device)
            return (t if t.dtype == torch.int64 else (t != 0).to(torch.int64)).contiguous()
        lab = labels.to(torch.int64).contiguous()  # labels keep their VALUES (token ids, -100) whatever the int dtype
        keep = [lab] + [i64(m) for m in masks] + ([i64(speech_mask)] if speech_mask is not None else [])
        rows = torch.empty(B * T, dtype=torch.int64, device=labels.device)
        row_labels = torch.empty(B * T, dtype=torch.int64, device=labels.device)
        meta = torch.empty(2, dtype=torch.int32, device=labels.device)
        check(load_lib().sd_loss_rows(lab.data_ptr(), _p(keep[-1] if speech_mask is not None else None),
                                      _p(keep[1] if masks else None), _p(keep[2] if len(masks) > 1 else None),
                                      rows.data_ptr(), row_labels.data_ptr(), meta.data_ptr(), B, T, _stream()), "sd_loss_rows")
        n_rows, bad = meta.tolist()
        if bad:
            raise ValueError("attention_mask is not right-padded (a 1 follows a 0): the HIP attention kernels take a "
                             "valid-prefix length per sequence, as ProcessedDataCollator produces (data.py:280-327)")
        return rows[:n_rows], row_labels[:n_rows]
    nxt = torch.full_like(labels, -100)
    nxt[:, :-1] = labels[:, 1:]
    valid = nxt != -100
    if speech_mask is not None:
        m = torch.zeros_like(valid)
        m[:, :-1] = speech_mask.to(labels.device)[:, 1:] != 0
        valid &= m
    flat = valid.reshape(-1)
    if labels.is_cuda:
        host = torch.stack([flat.sum()] + [left_padded(m.to(labels.device)).to(torch.int64) for m in masks]).tolist()
        if any(host[1:]):
            raise ValueError("attention_mask is not right-padded (a 1 follows a 0): the HIP attention kernels take a "
                             "valid-prefix length per sequence, as ProcessedDataCollator produces (data.py:280-327)")
        try:
            rows = torch.nonzero_static(flat, size=int(host[0])).reshape(-1)
        except (NotImplementedError, RuntimeError):
            rows = torch.nonzero(flat).reshape(-1)
    else:
        if any(bool(left_padded(m)) for m in masks):
            raise ValueError("attention_mask is not right-padded")
        rows = torch.nonzero(flat).reshape(-1)
    return rows, nxt.reshape(-1)[rows]


def colsum_reduce_batch(problems):
    """[(partials [nb, stride] fp32, out [H] bf16, H, accumulate), ...] (at most 8): out[c] (+)= sum_r partials[r, c] for
    every problem in ONE launch (sd_colsum_reduce_batch: a layer's four gain gradients)."""
    from ._lib import ColsumProblem
    n = len(problems)
    arr = (ColsumProblem * n)()
    for i, (part, out, H, acc) in enumerate(problems):
        _need(out, torch.bfloat16, "out")
        if not part.is_cuda or part.dtype != torch.float32 or part.dim() != 2 or part.stride(1) != 1:
            raise ValueError("partials: fp32 GPU matrix [nb, >= H] with unit column stride")
        arr[i] = ColsumProblem(part.data_ptr(), out.data_ptr(), part.shape[0], H, part.stride(0), int(acc))
    check(load_lib().sd_colsum_reduce_batch(C.cast(arr, C.c_void_p), n, _stream()), "sd_colsum_reduce_batch")


def rows_scatter(src, rows, M):
    src = _need(src, torch.bfloat16, "src")
    dst = torch.empty(M, src.shape[1], dtype=torch.bfloat16, device=src.device)
    check(load_lib().sd_rows_scatter(src.data_ptr(), rows.data_ptr(), dst.data_ptr(), src.shape[0], M, src.shape[1],
                                     _stream()), "sd_rows_scatter")
    return dst


# --------------------------------------------------------------------------------------- optimizer
def sumsq(x, out, partials=None):
    """out[0] += sum(x^2) (deterministic two-launch reduction).  ``partials``: fp32 scratch of SUMSQ_PARTIALS elements the
    caller owns (FlatAdamW keeps one); allocated per call when omitted, so concurrent calls never share state."""
    from ._lib import SUMSQ_PARTIALS
    if partials is None:
        partials = torch.empty(SUMSQ_PARTIALS, dtype=torch.float32, device=x.device)
    elif partials.dtype != torch.float32 or partials.numel() < SUMSQ_PARTIALS or not partials.is_cuda:
        raise ValueError(f"partials: fp32 GPU scratch of at least {SUMSQ_PARTIALS} elements")
    check(load_lib().sd_sumsq_bf16(x.data_ptr(), x.numel(), out.data_ptr(), partials.data_ptr(), _stream()),
          "sd_sumsq_bf16")


def adamw_(p, g, m, v, lr, beta1, beta2, eps, wd, step, grad_sumsq=None, max_norm=0.0):
    check(load_lib().sd_adamw_bf16(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2,
                                   eps, wd, step, _p(grad_sumsq), max_norm, _stream()), "sd_adamw_bf16")


def adamw_f32_shadow_(p, g, m, v, shadow, shadow_scaled, scale, lr, beta1, beta2, eps, wd, step, grad_sumsq=None,
                       max_norm=0.0):
    """AdamW on fp32 masters with a bf16 gradient; also writes shadow = bf16(p), shadow_scaled = bf16(scale p)."""
    check(load_lib().sd_adamw_f32_shadow(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), shadow.data_ptr(),
                                         shadow_scaled.data_ptr(), scale, p.numel(), lr, beta1, beta2, eps, wd, step,
                                         _p(grad_sumsq), max_norm, _stream()), "sd_adamw_f32_shadow")


# --------------------------------------------------------------------------------------- profiling
def prof_begin():
    check(load_lib().sd_prof_begin(), "sd_prof_begin")


def prof_end():
    """-> {kind: (total_ms, total_work, launches)} measured with HIP events on the launch stream."""
    from ._lib import KINDS
    n = len(KINDS)
    ms, work, cnt = (C.c_double * n)(), (C.c_double * n)(), (C.c_int64 * n)()
    check(load_lib().sd_prof_end(ms, work, cnt, n), "sd_prof_end")
    return {KINDS[i]: (ms[i], work[i], cnt[i]) for i in range(n)}


def prof_symbols():
    """-> {kernel symbol: (total_ms, total_work, launches, kind)} of the last prof_end (see sd_prof_symbols)."""
    from ._lib import KINDS
    lib = load_lib()
    n = lib.sd_prof_symbols(None, 0)
    buf = C.create_string_buffer(int(n))
    lib.sd_prof_symbols(buf, n)
    out = {}
    for line in buf.value.decode().splitlines():
        sym, kind, ms, work, cnt = line.split("\t")
        out[sym] = (float(ms), float(work), int(cnt), KINDS[int(kind)] if int(kind) < len(KINDS) else "?")
    return out


def rmsnorm_bwd_from_splitk(a, b_kn, x, w, rstd, dres=None):
    """dx, dw of RMSNorm where dy = a @ b_kn is produced by a split-K GEMM whose fp32 slabs the norm kernel sums
    itself (sd_gemm_bf16_splitk_partial + sd_rmsnorm_bwd_slabs).  Returns (dx, dw, nsplit)."""
    lib = load_lib()
    M, K = a.shape
    N = b_kn.shape[1]
    nb = max(lib.sd_gemm_splitk_workspace_bytes(M, N, K), 16)
    ws = torch.empty(nb, dtype=torch.uint8, device=a.device)
    dy = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    nsp = C.c_int(0)
    check(lib.sd_gemm_bf16_splitk_partial(a.data_ptr(), b_kn.data_ptr(), dy.data_ptr(), M, N, K, a.stride(0), b_kn.stride(0),
                                          N, 0, 1, ws.data_ptr(), nb, C.byref(nsp), _stream()), "sd_gemm_bf16_splitk_partial")
    wsn = torch.empty(lib.sd_rmsnorm_bwd_workspace_bytes(M, N), dtype=torch.uint8, device=a.device)
    dx, dw = torch.empty_like(x), torch.zeros_like(w)
    if nsp.value > 1:
        check(lib.sd_rmsnorm_bwd_slabs(ws.data_ptr(), nsp.value, x.data_ptr(), w.data_ptr(), rstd.data_ptr(), _p(dres),
                                       dx.data_ptr(), dw.data_ptr(), 0, wsn.data_ptr(), M, N, 0, 0, _stream()),
              "sd_rmsnorm_bwd_slabs")
    else:
        check(lib.sd_rmsnorm_bwd2(dy.data_ptr(), x.data_ptr(), w.data_ptr(), rstd.data_ptr(), _p(dres), dx.data_ptr(),
                                  dw.data_ptr(), 0, wsn.data_ptr(), M, N, 0, 0, _stream()), "sd_rmsnorm_bwd2")
    return dx, dw, nsp.value


# --------------------------------------------------------------------------------- fused forward GEMMs
def gemm_swiglu(x, wgu, save_gu=True):
    """act = silu(x Wg^T) * (x Wu^T) with wgu = [gate rows | up rows]; returns (act, gu or None)."""
    M, K = x.shape
    I = wgu.shape[0] // 2
    act = torch.empty(M, I, dtype=torch.bfloat16, device=x.device)
    gu = torch.empty(M, 2 * I, dtype=torch.bfloat16, device=x.device) if save_gu else None
    check(load_lib().sd_gemm_swiglu(x.data_ptr(), wgu.data_ptr(), _p(gu), act.data_ptr(), M, I, K, _stream()), "sd_gemm_swiglu")
    return act, gu


def gemm_qkv_rope(x, wqkv, q_gain, k_gain, cos, sin, T, Hq, Hkv, eps=1e-6):
    """raw q|k|v and RMS-normalised + RoPE-rotated q|k in one launch."""
    M, K = x.shape
    qkv = torch.empty(M, (Hq + 2 * Hkv) * 128, dtype=torch.bfloat16, device=x.device)
    qk = torch.empty(M, (Hq + Hkv) * 128, dtype=torch.bfloat16, device=x.device)
    check(load_lib().sd_gemm_qkv_rope(x.data_ptr(), wqkv.data_ptr(), qkv.data_ptr(), qk.data_ptr(), q_gain.data_ptr(),
                                      k_gain.data_ptr(), cos.data_ptr(), sin.data_ptr(), M, T, Hq, Hkv, K, eps, _stream()),
          "sd_gemm_qkv_rope")
    return qkv, qk


# ---- RMSNorm folded into the projection behind it (include/sd_hip.h "RMSNorm folded ..."; the frozen teacher)
def embedding_fwd_ssq(ids, E):
    ids = _need(ids, torch.int64, "ids").reshape(-1)
    V, H = E.shape
    x = torch.empty(ids.numel(), H, dtype=torch.bfloat16, device=E.device)
    ssq = torch.empty(H // 128, ids.numel(), dtype=torch.float32, device=E.device)  # tile-major
    check(load_lib().sd_embedding_fwd_ssq(ids.data_ptr(), _p(E), x.data_ptr(), ssq.data_ptr(), ids.numel(), H, V, _stream()),
          "sd_embedding_fwd_ssq")
    return x, ssq


def gemm_resid_ssq(a, w, residual):
    """C = a . w^T + residual and the per-128-column-tile sums of squares of C: ([M,N] bf16, [N/128,M] fp32 tile-major)."""
    M, K = a.shape
    N = w.shape[0]
    c = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    ssq = torch.empty(N // 128, M, dtype=torch.float32, device=a.device)
    check(load_lib().sd_gemm_bf16_ssq(_p(a), _p(w), c.data_ptr(), _p(residual), ssq.data_ptr(), M, N, K, a.stride(0),
                                      w.stride(0), N, residual.stride(0), _stream()), "sd_gemm_bf16_ssq")
    return c, ssq


def gemm_swiglu_rs(x, wgu_folded, ssq, eps=1e-6, save_gu=False):
    M, K = x.shape
    I = wgu_folded.shape[0] // 2
    act = torch.empty(M, I, dtype=torch.bfloat16, device=x.device)
    gu = torch.empty(M, 2 * I, dtype=torch.bfloat16, device=x.device) if save_gu else None
    check(load_lib().sd_gemm_swiglu_rs(_p(x), _p(wgu_folded), _p(gu), act.data_ptr(), _p(ssq), eps, M, I, K, _stream()),
          "sd_gemm_swiglu_rs")
    return act, gu


def gemm_qkv_rope_rs(x, wqkv_folded, q_gain, k_gain, cos, sin, ssq, T, Hq, Hkv, eps=1e-6):
    M, K = x.shape
    qkv = torch.empty(M, (Hq + 2 * Hkv) * 128, dtype=torch.bfloat16, device=x.device)
    qk = torch.empty(M, (Hq + Hkv) * 128, dtype=torch.bfloat16, device=x.device)
    check(load_lib().sd_gemm_qkv_rope_rs(_p(x), _p(wqkv_folded), qkv.data_ptr(), qk.data_ptr(), _p(q_gain), _p(k_gain),
                                         _p(cos), _p(sin), _p(ssq), M, T, Hq, Hkv, K, eps, _stream()), "sd_gemm_qkv_rope_rs")
    return qkv, qk
