"""Oracle: the optional LoRA student of train.py:180-202 (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

PARITY UNPINNED: the algorithm lives in the third-party package ``peft`` (requirements.txt:15, no version pinned),
which is neither vendored under /root/reference nor installed in the build image, and the reference holds no test or
golden vector for this path.  What follows restates peft's published LoRA layer (``peft/tuners/lora/layer.py``:
``LoraLayer.update_layer`` / ``reset_lora_parameters`` / ``pissa_init`` / ``Linear.forward``) as the reference
configures it at train.py:182-201:

  target_modules   q/k/v/o/gate/up/down_proj of every decoder layer       train.py:185-193
  modules_to_save  embed_tokens, lm_head: full trainable copies, made     train.py:194
                   independently of each other -> a TIED pair trains UNTIED
  r = 32, lora_alpha = 64, lora_dropout = 0, bias = "none"                train.py:183-184, 195-196, :474-475
  use_rslora       scaling = alpha / sqrt(r)   (else alpha / r)           train.py:198, :476-481
  init             "pissa" (default) | "pissa_niter_N" | "gaussian" | True train.py:199, :482-487

  y = x W^T + scaling * (x A^T) B^T            A [r, in], B [out, r]

  True / "default":  A ~ kaiming_uniform(a = sqrt 5) = U(-1/sqrt(in), 1/sqrt(in)),  B = 0
  "gaussian":        A ~ N(0, (1/r)^2),                                              B = 0
  "pissa":           W = U S V^T (fp32);  A = diag(sqrt(S_r / scaling)) V_r^T,  B = U_r diag(sqrt(S_r / scaling)),
                     W <- W - scaling * B A   (the frozen base keeps the residual; stored back in the base dtype)

peft keeps A and B in fp32 next to a bf16 base model (``autocast_adapter_dtype=True``): they are fp32 masters here too.
Because y is linear in W, the adapter is evaluated on the MERGED weight W + scaling * B A -- the same function, and the
form the HIP path uses (one rank-r update of the weight per optimizer step instead of two skinny GEMMs per projection
per micro-batch).
"""
from __future__ import annotations

import math

import torch

from . import qwen3 as Q
from . import step as S

TARGETS = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")


def scaling(r, lora_alpha, use_rslora=True):
    return lora_alpha / math.sqrt(r) if use_rslora else lora_alpha / r


def target_names(shape, targets=TARGETS):
    out = []
    for l in range(shape.num_hidden_layers):
        for t in TARGETS:  # fixed order
            if t in targets:
                grp = "mlp" if t in ("gate_proj", "up_proj", "down_proj") else "self_attn"
                out.append(f"model.layers.{l}.{grp}.{t}.weight")
    return out


def init_pair(W, r, scale, method, gen):
    """(A [r,in] fp32, B [out,r] fp32, residual base weight fp32) for one target of weight W [out,in]."""
    W = W.float()
    out_f, in_f = W.shape
    if method in (True, "default", "true", "True"):
        bound = 1.0 / math.sqrt(in_f)
        A = (torch.rand(r, in_f, generator=gen) * 2 - 1) * bound
        return A, torch.zeros(out_f, r), W
    if method == "gaussian":
        return torch.randn(r, in_f, generator=gen) / r, torch.zeros(out_f, r), W
    if method == "pissa":
        U, Sv, Vh = torch.linalg.svd(W, full_matrices=False)
        Ur, Sr, Vhr = U[:, :r], Sv[:r] / scale, Vh[:r]
    elif isinstance(method, str) and method.startswith("pissa_niter_"):
        Ur, Sr, Vr = torch.svd_lowrank(W, r, niter=int(method.split("_niter_")[-1]))
        Sr, Vhr = Sr / scale, Vr.t()
    else:
        raise ValueError(f"init_lora_weights={method!r}")
    A = torch.diag(torch.sqrt(Sr)) @ Vhr
    B = Ur @ torch.diag(torch.sqrt(Sr))
    return A, B, W - scale * (B @ A)


def attach(student_w, shape, r=32, lora_alpha=64, use_rslora=True, init="pissa", seed=0, targets=TARGETS,
           base_dtype=torch.bfloat16):
    """Returns (base_w, lora) -- base_w: the frozen weights (targets hold the residual, rounded through ``base_dtype``
    as peft stores it back into the bf16 model), plus an untied ``lm_head.weight`` copy (modules_to_save);
    lora: name -> (A, B)."""
    s = scaling(r, lora_alpha, use_rslora)
    gen = torch.Generator().manual_seed(seed)
    base = {k: v.clone() for k, v in student_w.items()}
    if "lm_head.weight" not in base:
        base["lm_head.weight"] = base["model.embed_tokens.weight"].clone()
    lora = {}
    for name in target_names(shape, targets):
        A, B, res = init_pair(base[name], r, s, init, gen)
        base[name] = res.to(base_dtype).to(student_w[name].dtype) if base_dtype is not None else res
        lora[name] = (A, B)
    return base, lora


def merged_weights(base_w, lora, scale, storage=None):
    """name -> W + scale * B A for targets (autograd flows to A and B), the frozen tensor otherwise.
    storage="bf16": what the HIP path keeps in bf16 on this path -- the operands bf16(A), bf16(scale * B) of the merge,
    the merged weight, and (backward of the same roundings) the weight gradient dW and the projected dA / dB."""
    w = dict(base_w)
    for name, (A, B) in lora.items():
        w[name] = Q._st(base_w[name] + Q._st(scale * B, storage) @ Q._st(A, storage), storage)
    return w


def lora_step(base_w, lora, shape, teacher_w, teacher_shape, batch, r=32, lora_alpha=64, use_rslora=True,
              storage=None, **kw):
    """One compute_loss + backward of the LoRA student (autograd from the A / B / embed / head leaves).  Returns
    distill_step's dict with ``grads`` = {"<target module>.lora_A": dA, "....lora_B": dB,
    "model.embed_tokens.weight": ..., "lm_head.weight": ...}."""
    s = scaling(r, lora_alpha, use_rslora)
    untied = Q.Qwen3Shape(**{**shape.__dict__, "tie_word_embeddings": False})
    leaves, pairs = {}, {}
    for n, (A, B) in lora.items():
        a, b = A.detach().clone().requires_grad_(True), B.detach().clone().requires_grad_(True)
        pairs[n] = (a, b)
        leaves[n[:-len("weight")] + "lora_A"], leaves[n[:-len("weight")] + "lora_B"] = a, b
    frozen = {k: v.detach() for k, v in base_w.items()}
    for k in ("model.embed_tokens.weight", "lm_head.weight"):
        frozen[k] = leaves[k] = frozen[k].clone().requires_grad_(True)
    w = merged_weights(frozen, pairs, s, storage)
    return S.distill_step(w, untied, teacher_w, teacher_shape, batch, storage=storage, grad_wrt=leaves, **kw)
