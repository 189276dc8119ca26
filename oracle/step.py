"""Oracle: one Stage-2 distillation micro-step, restated from train.py:43-116.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

  student forward ............................... train.py:54
  teacher no-grad forward (skipped when the batch
    already carries pre-extracted top-K) ........ train.py:60-69
  on-the-fly log-softmax + top-K (unless the
    teacher is quantized or top_k <= 0) ......... train.py:74-94
  DistillationLoss .............................. train.py:97-104
Also the collator's label rule (data.py:219-278): labels = ids, pad -> -100, every
position before the first speech_bos -> -100 (rows without speech_bos: all -100).
"""
from __future__ import annotations

import torch

from . import distill_loss as L
from . import qwen3 as Q


def make_labels(input_ids, pad_token_id, speech_bos_id):
    """data.py:245-276 without the per-row Python loop."""
    labels = input_ids.clone()
    labels[labels == pad_token_id] = -100
    is_bos = input_ids == speech_bos_id
    seen = is_bos.cumsum(-1) > 0
    labels[~seen] = -100
    return labels


def collate(features, pad_token_id, speech_bos_id, pad_to_multiple_of=None):
    """ProcessedDataCollator.__call__ restated (data.py:219-348): right padding."""
    def pad(seqs, masks):
        n = max(len(s) for s in seqs)
        if pad_to_multiple_of:
            n = (n + pad_to_multiple_of - 1) // pad_to_multiple_of * pad_to_multiple_of
        ids = torch.full((len(seqs), n), pad_token_id, dtype=torch.long)
        am = torch.zeros((len(seqs), n), dtype=torch.long)
        for i, (s, m) in enumerate(zip(seqs, masks)):
            ids[i, : len(s)] = torch.as_tensor(s, dtype=torch.long)
            am[i, : len(m)] = torch.as_tensor(m, dtype=torch.long)
        return ids, am

    if "student_input_ids" in features[0]:
        ids, am = pad([f["student_input_ids"] for f in features], [f["student_attention_mask"] for f in features])
    else:
        ids, am = pad([f["input_ids"] for f in features], [f["attention_mask"] for f in features])
    batch = {"input_ids": ids, "attention_mask": am}
    batch["labels"] = make_labels(ids, pad_token_id, speech_bos_id)
    if features[0].get("teacher_input_ids") is not None:
        tids, tam = pad([f["teacher_input_ids"] for f in features], [f["teacher_attention_mask"] for f in features])
        batch["teacher_input_ids"], batch["teacher_attention_mask"] = tids, tam
    if features[0].get("teacher_top_k_v") is not None:
        n = ids.shape[1]

        def padk(key, dtype):
            out = []
            for f in features:
                t = torch.as_tensor(f[key]).to(dtype)
                if t.shape[0] < n:
                    t = torch.cat([t, torch.zeros(n - t.shape[0], t.shape[1], dtype=dtype)], 0)
                out.append(t[:n])
            return torch.stack(out)

        batch["teacher_top_k_v"] = padk("teacher_top_k_v", torch.float16)
        batch["teacher_top_k_i"] = padk("teacher_top_k_i", torch.int32)
    return batch


def distill_step(student_w, student_shape, teacher_w, teacher_shape, batch, temperature=2.0, alpha=0.5,
                 top_k=100, is_quantized_teacher=False, with_grad=True, acc=torch.float64, storage=None, grad_wrt=None):
    """One compute_loss (+ backward).  Returns dict(total, task, distill, teacher, grads, logits).

    storage="bf16": the error-budget mode of oracle/qwen3.py -- weights rounded to bf16 (the HIP model keeps bf16
    parameters, train.py:174), activations and their gradients rounded at the HIP path's storage points, parameter
    gradients rounded to bf16 at the end (the flat bf16 gradient buffer); the loss itself stays fp32 on the (rounded)
    logits, as in the HIP kernels.
    """
    if storage == "bf16":
        student_w = {k: v.to(torch.bfloat16).to(v.dtype) for k, v in student_w.items()}
        if teacher_w is not None:
            teacher_w = {k: v.to(torch.bfloat16).to(v.dtype) for k, v in teacher_w.items()}
    if grad_wrt is None:
        sw = {k: v.detach().clone().requires_grad_(with_grad) for k, v in student_w.items()}
        wrt = sw
    else:   # student_w is built (differentiably) from the leaves in grad_wrt: oracle/lora.py
        sw, wrt = student_w, grad_wrt
    logits = Q.forward(sw, student_shape, batch["input_ids"], batch.get("attention_mask"), storage=storage)
    tkv, tki = batch.get("teacher_top_k_v"), batch.get("teacher_top_k_i")
    t_logits = None
    if tkv is None and teacher_w is not None:
        with torch.no_grad():
            if batch.get("teacher_input_ids") is not None:
                t_logits = Q.forward(teacher_w, teacher_shape, batch["teacher_input_ids"],
                                     batch.get("teacher_attention_mask"), storage=storage)
            else:
                t_logits = Q.forward(teacher_w, teacher_shape, batch["input_ids"], batch.get("attention_mask"),
                                     storage=storage)
    if t_logits is not None and tkv is None and not is_quantized_teacher and top_k > 0:
        tkv, tki = L.extract_topk(t_logits, top_k, vocab_size=logits.shape[-1])
        t_logits = None
    total, task, distill, teacher = L.distill_loss(
        logits, batch["labels"], teacher_logits=t_logits, teacher_top_k_v=tkv, teacher_top_k_i=tki,
        temperature=temperature, alpha=alpha, acc=acc)
    out = {"total": total.detach(), "task": task.detach(), "distill": distill.detach(),
           "teacher": teacher.detach(), "logits": logits.detach(), "top_k_v": tkv, "top_k_i": tki}
    if with_grad and total.requires_grad:
        total.backward()
        out["grads"] = {k: (v.grad.to(torch.bfloat16).to(v.grad.dtype) if storage == "bf16" else v.grad)
                        for k, v in wrt.items() if v.grad is not None}
    return out


def grad_error_budget(hip_grads, fp32_grads, bf16_grads, names=None):
    """Per tensor: relative L2 error of the HIP gradient and of the bf16-storage oracle's gradient, both against the
    fp32 oracle, and their ratio (assert ratio <= 1.5).  All arguments: dict name -> tensor."""
    out = {}
    for k in (names or fp32_grads.keys()):
        ref = fp32_grads[k].double().reshape(-1)
        rn = float(ref.norm())
        eh = float((hip_grads[k].detach().double().cpu().reshape(-1) - ref).norm()) / max(rn, 1e-300)
        eb = float((bf16_grads[k].double().reshape(-1) - ref).norm()) / max(rn, 1e-300)
        out[k] = {"err_hip": eh, "err_bf16_oracle": eb, "ratio": eh / max(eb, 1e-300)}
    return out
