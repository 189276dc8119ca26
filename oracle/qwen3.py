"""Oracle: Qwen3 causal-LM forward restated from scratch in plain torch (CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The arithmetic lives in a third-party dependency of the reference
(``transformers``, pinned ==4.57.1 in /root/reference/requirements.txt:9; 5.15.0
installed here), module ``transformers.models.qwen3.modeling_qwen3``; the
reference calls it at train.py:54 (student), train.py:63-69 (teacher) and
extract_teacher_logits.py:110.  Restated pieces and the HF lines they follow:
  RMSNorm (fp32 stats, cast, then * w) ........ modeling_qwen3.py:59-64
  SwiGLU MLP .................................. modeling_qwen3.py:81-83
  RoPE table (theta^(-2i/d), fp32, cat halves). modeling_qwen3.py:113-137
  rotate-half application ..................... modeling_qwen3.py:140-170
  GQA repeat (kv head j serves q heads j*g..).. modeling_qwen3.py:173-182
  softmax(QK^T d^-1/2 + mask) V, fp32 softmax . modeling_qwen3.py:185-207
  q_norm/k_norm over head_dim BEFORE RoPE ..... modeling_qwen3.py:252-257
  pre-norm residual decoder layer ............. modeling_qwen3.py:304-323
  embed -> layers -> final norm -> tied head .. modeling_qwen3.py:381-441
Shapes of the teacher: /root/reference/soulxpodcast/config.py:12-42.

Weights are a flat ``dict[str, Tensor]`` with HF state-dict key names, so the
same dict drives HF (when pinning the oracle), this oracle, and the HIP model.
Gradients are taken with torch autograd over this forward.

``storage="bf16"`` (round 4) is the ERROR-BUDGET mode: the same fp32 arithmetic, but every tensor
the HIP path keeps in HBM as bf16 (normalised rows, q|k|v, rotated q|k, softmax probabilities handed to the P.V
product, attention output, both residual sums, gate, up, act, final norm, logits) is rounded to bf16 where the HIP
path stores it -- and, through ``_RoundBF16.backward``, so is the gradient that flows back through the same point
(the HIP backward stores those gradients as bf16 too).  It is NOT a bit-exact emulation (summation orders differ);
it measures how much of |HIP - fp32| is plain bf16 storage noise, so that the GPU tests can assert
``err(HIP, fp32) <= 1.5 * err(bf16-storage oracle, fp32)`` per tensor instead of a bare cosine.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch


@dataclass
class Qwen3Shape:
    vocab_size: int
    hidden_size: int
    intermediate_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    head_dim: int = 128
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    tie_word_embeddings: bool = True

    @property
    def q_dim(self):
        return self.num_attention_heads * self.head_dim

    @property
    def kv_dim(self):
        return self.num_key_value_heads * self.head_dim


# real shapes
STUDENT_06B = Qwen3Shape(159488, 1024, 3072, 28, 16, 8)
TEACHER_17B = Qwen3Shape(159488, 2048, 6144, 28, 16, 8)


def param_names(shape: Qwen3Shape):
    """HF state-dict keys and shapes, in HF module order."""
    h, I, d = shape.hidden_size, shape.intermediate_size, shape.head_dim
    out = [("model.embed_tokens.weight", (shape.vocab_size, h))]
    for l in range(shape.num_hidden_layers):
        p = f"model.layers.{l}."
        out += [
            (p + "self_attn.q_proj.weight", (shape.q_dim, h)),
            (p + "self_attn.k_proj.weight", (shape.kv_dim, h)),
            (p + "self_attn.v_proj.weight", (shape.kv_dim, h)),
            (p + "self_attn.o_proj.weight", (h, shape.q_dim)),
            (p + "self_attn.q_norm.weight", (d,)),
            (p + "self_attn.k_norm.weight", (d,)),
            (p + "mlp.gate_proj.weight", (I, h)),
            (p + "mlp.up_proj.weight", (I, h)),
            (p + "mlp.down_proj.weight", (h, I)),
            (p + "input_layernorm.weight", (h,)),
            (p + "post_attention_layernorm.weight", (h,)),
        ]
    out.append(("model.norm.weight", (h,)))
    if not shape.tie_word_embeddings:
        out.append(("lm_head.weight", (shape.vocab_size, h)))
    return out


def init_weights(shape: Qwen3Shape, seed: int = 0, std: float = 0.02, norm_jitter: float = 0.0,
                 dtype=torch.float32):
    """Deterministic random init from a CPU generator (HF default: N(0, 0.02), norms = 1).

    ``norm_jitter`` > 0 perturbs the RMSNorm gains away from 1 so that parity
    tests can see a wrong/missing gain.
    """
    g = torch.Generator().manual_seed(seed)
    w = {}
    for name, shp in param_names(shape):
        if len(shp) == 1:
            t = torch.ones(shp)
            if norm_jitter:
                t = t + norm_jitter * torch.randn(shp, generator=g)
        else:
            t = torch.randn(shp, generator=g) * std
        w[name] = t.to(dtype)
    return w


class _RoundBF16(torch.autograd.Function):
    """y = bf16(x) in the forward, dx = bf16(dy) in the backward: one bf16 STORAGE point of the HIP path."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def _st(x, storage):
    return _RoundBF16.apply(x) if storage == "bf16" else x


def rms_norm(x, w, eps, storage=None):
    """modeling_qwen3.py:59-64: fp32 statistics, cast back, THEN multiply by the gain."""
    dt = x.dtype
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    # HF in bf16 rounds at the cast AND after the gain; the HIP kernels do the same (sd_elementwise.hip, EPI 4)
    return _st(w * _st(xf.to(dt), storage), storage)


def rope_tables(T, d, theta, dtype):
    """modeling_qwen3.py:113-137: inv_freq = theta^(-2i/d); emb = cat(freqs, freqs)."""
    inv = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    pos = torch.arange(T, dtype=torch.float32)
    fr = pos[:, None] * inv[None, :]
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def apply_rope(x, cos, sin):
    """modeling_qwen3.py:140-170 (rotate-half, non-interleaved). x: [B,H,T,d]."""
    d = x.shape[-1]
    x1, x2 = x[..., : d // 2], x[..., d // 2:]
    rot = torch.cat((-x2, x1), dim=-1)
    return x * cos[None, None] + rot * sin[None, None]


def attention(q, k, v, key_len=None, storage=None):
    """Causal GQA attention, fp32 softmax (modeling_qwen3.py:185-207).

    q [B,Hq,T,d], k/v [B,Hkv,T,d]; ``key_len`` [B] masks right padding keys.
    """
    B, Hq, T, d = q.shape
    g = Hq // k.shape[1]
    k = k.repeat_interleave(g, dim=1)
    v = v.repeat_interleave(g, dim=1)
    s = (q @ k.transpose(2, 3)) * (d ** -0.5)
    mask = torch.ones(T, T, dtype=torch.bool).tril()[None, None]
    if key_len is not None:
        kk = torch.arange(T)[None, :] < key_len[:, None]
        mask = mask & kk[:, None, None, :]
    s = s.masked_fill(~mask, torch.finfo(s.dtype).min)
    p = torch.softmax(s, dim=-1, dtype=torch.float32).to(q.dtype)
    return _st(p, storage) @ v  # flash kernels (and the HIP one) hand P to the matrix unit as bf16


def forward(w, shape: Qwen3Shape, input_ids, attention_mask=None, return_hidden=False, storage=None):
    """input_ids [B,T] -> logits [B,T,V] in the dtype of the weights.  storage="bf16": see the module docstring."""
    st = lambda t: _st(t, storage)  # noqa: E731
    B, T = input_ids.shape
    Hq, Hkv, d = shape.num_attention_heads, shape.num_key_value_heads, shape.head_dim
    eps = shape.rms_norm_eps
    emb = w["model.embed_tokens.weight"]
    x = emb[input_ids]
    cos, sin = rope_tables(T, d, shape.rope_theta, x.dtype)
    key_len = None if attention_mask is None else attention_mask.sum(-1)
    hiddens = [x]
    for l in range(shape.num_hidden_layers):
        p = f"model.layers.{l}."
        r = x
        xn = rms_norm(x, w[p + "input_layernorm.weight"], eps, storage)
        q = st(xn @ w[p + "self_attn.q_proj.weight"].T).view(B, T, Hq, d)
        k = st(xn @ w[p + "self_attn.k_proj.weight"].T).view(B, T, Hkv, d)
        v = st(xn @ w[p + "self_attn.v_proj.weight"].T).view(B, T, Hkv, d)
        q = rms_norm(q, w[p + "self_attn.q_norm.weight"], eps, storage).transpose(1, 2)
        k = rms_norm(k, w[p + "self_attn.k_norm.weight"], eps, storage).transpose(1, 2)
        v = v.transpose(1, 2)
        q = st(apply_rope(q, cos, sin))
        k = st(apply_rope(k, cos, sin))
        o = st(attention(q, k, v, key_len, storage)).transpose(1, 2).reshape(B, T, Hq * d)
        x = st(r + o @ w[p + "self_attn.o_proj.weight"].T)
        r = x
        xn = rms_norm(x, w[p + "post_attention_layernorm.weight"], eps, storage)
        gate = st(xn @ w[p + "mlp.gate_proj.weight"].T)
        up = st(xn @ w[p + "mlp.up_proj.weight"].T)
        x = st(r + st(torch.nn.functional.silu(gate) * up) @ w[p + "mlp.down_proj.weight"].T)
        hiddens.append(x)
    x = rms_norm(x, w["model.norm.weight"], eps, storage)
    head = emb if shape.tie_word_embeddings else w["lm_head.weight"]
    logits = st(x @ head.T)
    if return_hidden:
        return logits, hiddens
    return logits


def flops_per_token(shape: Qwen3Shape, T: int):
    """Algorithmic forward FLOPs/token: 2*matmul params + causal attention."""
    h, I = shape.hidden_size, shape.intermediate_size
    per_layer = h * (shape.q_dim + 2 * shape.kv_dim) + shape.q_dim * h + 3 * h * I
    params = shape.num_hidden_layers * per_layer + shape.vocab_size * h
    attn = 0.5 * 4 * T * shape.num_attention_heads * shape.head_dim * shape.num_hidden_layers
    return 2 * params + attn
