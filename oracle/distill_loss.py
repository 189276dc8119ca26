"""Oracle: temperature-scaled KL + CE distillation loss, restated from scratch.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows /root/reference/distillation_loss.py:
  * causal shift + valid-row predicate .......... distillation_loss.py:31-45
  * N == 0 guard ................................ distillation_loss.py:47-53
  * dense KL (batchmean, x T^2) + teacher CE .... distillation_loss.py:56-71
  * sparse top-K KL + approximate teacher CE .... distillation_loss.py:73-118
  * CE task loss + alpha mix .................... distillation_loss.py:123-128
and the on-the-fly extraction of /root/reference/train.py:74-94.

Everything is written as explicit per-row log-sum-exp arithmetic (no
``F.kl_div`` / ``F.cross_entropy`` / boolean row compaction) in a caller-chosen
accumulation dtype (fp64 by default) so that it is an independent statement of
the maths rather than a re-call of the same torch ops the reference uses.
"""
from __future__ import annotations

import torch

IGNORE = -100


def _rows(student_logits, labels, speech_token_mask=None):
    """Shifted views + validity predicate (distillation_loss.py:31-41)."""
    V = student_logits.shape[-1]
    s = student_logits[..., :-1, :].reshape(-1, V)
    y = labels[..., 1:].reshape(-1)
    valid = y != IGNORE
    if speech_token_mask is not None:
        valid = valid & speech_token_mask[..., 1:].reshape(-1).bool()
    return s, y, valid


def _lse(x):
    m = x.max(dim=-1, keepdim=True).values
    return (m + (x - m).exp().sum(dim=-1, keepdim=True).log()).squeeze(-1)


def distill_loss(
    student_logits,
    labels,
    teacher_logits=None,
    teacher_top_k_v=None,
    teacher_top_k_i=None,
    speech_token_mask=None,
    temperature: float = 2.0,
    alpha: float = 0.5,
    acc=torch.float64,
    return_grad: bool = False,
):
    """Returns (total, task, distill, teacher_task) as ``acc``-dtype 0-d tensors.

    With ``return_grad`` also returns d total / d student_logits ([B,T,V], acc
    dtype) from the closed form of SURVEY.md section 8a row L-7:
        dense : a*(softmax(s) - e_y)/N + (1-a)*T*(softmax(s/T) - q)/N
        sparse: same with q scattered (summed over duplicates) at the K indices.
    """
    T = float(temperature)
    a = float(alpha)
    B, S, V = student_logits.shape
    s_all, y_all, valid = _rows(student_logits, labels, speech_token_mask)
    N = int(valid.sum())
    zero = torch.zeros((), dtype=acc)
    if N == 0:  # distillation_loss.py:47-53 (constant zeros)
        out = (zero, zero.clone(), zero.clone(), zero.clone())
        if return_grad:
            return out + (torch.zeros(B, S, V, dtype=acc),)
        return out

    if teacher_logits is None and (teacher_top_k_v is None or teacher_top_k_i is None):
        raise ValueError("Either teacher_logits or top_k must be provided")

    rows = valid.nonzero(as_tuple=True)[0]
    s = s_all[rows].to(acc)
    y = y_all[rows].long()
    ar = torch.arange(N)

    lse1 = _lse(s)
    lseT = _lse(s / T)
    task = (lse1 - s[ar, y]).mean()  # distillation_loss.py:123

    g = None
    if return_grad:
        p1 = (s - lse1[:, None]).exp()
        p1[ar, y] -= 1.0
        pT = (s / T - lseT[:, None]).exp()
        g = a * p1 / N + (1.0 - a) * T * pT / N

    if teacher_logits is not None:
        t = teacher_logits[..., :-1, :].reshape(-1, teacher_logits.shape[-1])[rows].to(acc)
        t_lseT = _lse(t / T)
        logq = t / T - t_lseT[:, None]
        q = logq.exp()
        logp = s / T - lseT[:, None]
        # nn.KLDivLoss("batchmean"): sum_{n,v} q (log q - log p) / N, 0 log 0 := 0
        term = torch.where(q > 0, q * (logq - logp), torch.zeros_like(q))
        distill = term.sum() / N * (T * T)
        teacher_task = (_lse(t) - t[ar, y]).mean()  # distillation_loss.py:71
        if return_grad:
            g = g - (1.0 - a) * T * q / N
    else:
        K = teacher_top_k_v.shape[-1]
        v = teacher_top_k_v[..., :-1, :].reshape(-1, K)[rows].to(torch.float32).to(acc)
        idx = teacher_top_k_i[..., :-1, :].reshape(-1, K)[rows].long()
        v_lseT = _lse(v / T)
        logq = v / T - v_lseT[:, None]
        q = logq.exp()
        logp_g = s.gather(-1, idx) / T - lseT[:, None]
        distill = (q * (logq - logp_g)).sum(-1).mean() * (T * T)
        hit = idx == y[:, None]  # distillation_loss.py:110-118
        if bool(hit.any()):
            teacher_task = -(v[hit]).mean()
        else:
            teacher_task = zero.clone()
        if return_grad:
            g = g.scatter_add(-1, idx, -(1.0 - a) * T * q / N)

    total = a * task + (1.0 - a) * distill
    out = (total, task, distill, teacher_task)
    if return_grad:
        full = torch.zeros(B * (S - 1), V, dtype=acc)
        full[rows] = g
        grad = torch.zeros(B, S, V, dtype=acc)
        grad[:, :-1, :] = full.view(B, S - 1, V)
        return out + (grad,)
    return out


def extract_topk(teacher_logits, k: int, vocab_size: int | None = None, acc=torch.float32):
    """On-the-fly sparse extraction (train.py:80-91; extract_teacher_logits.py:114-129).

    logits -> truncate to ``vocab_size`` -> log_softmax at T=1 -> top-k sorted
    descending -> (values fp16, indices int32).  Ties are broken towards the
    LOWEST index (torch.topk leaves tie order unspecified; the HIP kernel and
    this oracle both define it this way).
    """
    x = teacher_logits if vocab_size is None else teacher_logits[..., :vocab_size]
    x = x.to(acc)
    logp = x - _lse(x)[..., None]
    V = logp.shape[-1]
    # stable descending sort == ties towards the lowest index
    order = torch.sort(logp, dim=-1, descending=True, stable=True).indices[..., :k]
    vals = logp.gather(-1, order)
    assert order.max() < V
    return vals.to(torch.float16), order.to(torch.int32)
