"""Qwen3 causal LM (student 0.6B-shape / teacher SoulX-Podcast-1.7B-shape) on the HIP step runner.

Model protocol the reference relies on: callable
``model(input_ids=, attention_mask=, labels=?, **kw)`` returning an object with ``.logits`` [B,T,V]
(train.py:54-55, 63-69); ``.eval()``, ``.parameters()``, ``requires_grad_`` (train.py:165-169);
``gradient_checkpointing_enable()`` / ``enable_input_require_grads()`` (train.py:206-208); ``.config``;
``state_dict`` with HF key names.  The forward keeps ``**kwargs`` so that HF Trainer's
``model_accepts_loss_kwargs`` stays True exactly as for HF Qwen3 (quirk Q1).

Layout in HBM: ONE flat bf16 parameter buffer and ONE flat bf16 gradient buffer, ordered
[embed | layer 0 .. layer L-1 | final norm]; inside a layer q|k|v projection rows are contiguous
(one fused [4096,h] GEMM) and so are gate|up.  HF-named ``nn.Parameter``s are views into the flat
buffers, so a data-parallel all-reduce bucket is a plain slice.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, load_lib
from .ops import _need, _p, _stream, left_padded, rope_tables  # noqa: F401  (left_padded is re-exported)


@dataclass
class Qwen3Dims:
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

    @classmethod
    def from_hf_config(cls, c):
        """From an HF Qwen3Config (transformers 4.x keeps ``rope_theta`` on the config, 5.x in ``rope_parameters``)."""
        rope = getattr(c, "rope_theta", None)
        if rope is None:
            rope = (getattr(c, "rope_parameters", None) or {}).get("rope_theta", 1e6)
        return cls(c.vocab_size, c.hidden_size, c.intermediate_size, c.num_hidden_layers, c.num_attention_heads,
                   c.num_key_value_heads, getattr(c, "head_dim", None) or 128, c.rms_norm_eps, float(rope),
                   bool(c.tie_word_embeddings))

    @classmethod
    def student_06b(cls):  # public Qwen3-0.6B shape, vocab expanded to the teacher's (prepare_student.py:31,79-82)
        return cls(159488, 1024, 3072, 28, 16, 8)

    @classmethod
    def teacher_17b(cls):  # soulxpodcast/config.py:12-42
        return cls(159488, 2048, 6144, 28, 16, 8)

    def matmul_params(self):
        """Parameters that take part in a matmul per token: the decoder's projections + the (tied) lm_head."""
        h, I = self.hidden_size, self.intermediate_size
        per_layer = h * (self.q_dim + 2 * self.kv_dim) + self.q_dim * h + 3 * h * I
        return self.num_hidden_layers * per_layer + self.vocab_size * h

    def flops_per_token(self, T):
        """Algorithmic forward FLOPs per token: 2 per matmul parameter + causal attention
        (half of 4*T*Hq*d per layer).  Backward = 2x this; a distillation micro-step = 3x student + 1x teacher."""
        attn = 0.5 * 4 * T * self.num_attention_heads * self.head_dim * self.num_hidden_layers
        return 2 * self.matmul_params() + attn

    def lm_head_flops_per_row(self):
        return 2.0 * self.hidden_size * self.vocab_size


class _Holder(nn.Module):
    """Leaf module owning one HF-named ``weight`` Parameter (a view into the flat buffer)."""

    def __init__(self, w):
        super().__init__()
        self.weight = w


class CausalLMOutput(dict):
    """Minimal ModelOutput: attribute + key access, like HF's CausalLMOutputWithPast."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


SAVE_NONE, SAVE_ALL, SAVE_LAYER_INPUTS, SAVE_NONE_FOLDED = 0, 1, 2, 3  # include/sd_hip.h SD_SAVE_*
FWD_CONCURRENT = 0x100                            # SD_FWD_CONCURRENT
BWD_ACCUMULATE, BWD_RECOMPUTE = 1, 2              # SD_BWD_*


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, input_ids, kv_len, model, rows, concurrent=False):
        save = SAVE_LAYER_INPUTS if model._wants_recompute(*input_ids.shape, input_ids.device) else SAVE_ALL
        logits, acts = model._run_forward(input_ids, kv_len, save=save, rows=rows, concurrent=concurrent)
        ctx.model, ctx.acts, ctx.ids, ctx.kv_len, ctx.rows, ctx.save = model, acts, input_ids, kv_len, rows, save
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        ctx.model._run_backward(ctx.ids, ctx.kv_len, ctx.acts, dlogits, rows=ctx.rows,
                                recompute=ctx.save == SAVE_LAYER_INPUTS)
        ctx.acts = None
        return torch.zeros((), device=dlogits.device), None, None, None, None, None


class HipQwen3ForCausalLM(nn.Module):
    def __init__(self, dims: Qwen3Dims, device="cuda", config=None, init_std=0.02, seed=0):
        super().__init__()
        if dims.head_dim != 128:
            raise ValueError("the gfx950 attention / RoPE kernels are specialised for head_dim 128")
        self.dims = dims
        self.config = config if config is not None else self._hf_config(dims)
        self.gradient_checkpointing = False
        self.recompute_policy, self.recompute_fraction = "auto", 0.25
        d = dims
        h, I, qd, kd = d.hidden_size, d.intermediate_size, d.q_dim, d.kv_dim
        # ---- flat layout
        self._slices = {}
        off = 0

        def take(name, shape):
            nonlocal off
            n = 1
            for s in shape:
                n *= s
            self._slices[name] = (off, n, shape)
            off += (n + 7) // 8 * 8  # keep every tensor 16-byte aligned
        take("model.embed_tokens.weight", (d.vocab_size, h))
        self.layer_ranges = []
        for l in range(d.num_hidden_layers):
            p = f"model.layers.{l}."
            start = off
            take(p + "self_attn.q_proj.weight", (qd, h))
            take(p + "self_attn.k_proj.weight", (kd, h))
            take(p + "self_attn.v_proj.weight", (kd, h))
            take(p + "self_attn.o_proj.weight", (h, qd))
            take(p + "mlp.gate_proj.weight", (I, h))
            take(p + "mlp.up_proj.weight", (I, h))
            take(p + "mlp.down_proj.weight", (h, I))
            take(p + "self_attn.q_norm.weight", (128,))
            take(p + "self_attn.k_norm.weight", (128,))
            take(p + "input_layernorm.weight", (h,))
            take(p + "post_attention_layernorm.weight", (h,))
            self.layer_ranges.append((start, off))
        self.norm_range = (off, off + h)
        take("model.norm.weight", (h,))
        if not d.tie_word_embeddings:
            take("lm_head.weight", (d.vocab_size, h))
        self.numel_flat = off
        self.embed_range = (0, self._slices["model.embed_tokens.weight"][1])
        dev = torch.device(device)
        self.flat = torch.zeros(off, dtype=torch.bfloat16, device=dev)
        self.flat_grad = None
        self._grads_live = False
        # ---- HF-named parameters (views)
        self.model = nn.Module()
        self.model.layers = nn.ModuleList()
        self._params = {}
        for name, (o, n, shape) in self._slices.items():
            self._params[name] = nn.Parameter(self.flat[o:o + n].view(shape))
        self.model.embed_tokens = _Holder(self._params["model.embed_tokens.weight"])
        for l in range(d.num_hidden_layers):
            p = f"model.layers.{l}."
            lay = nn.Module()
            lay.self_attn = nn.Module()
            for nm in ("q_proj", "k_proj", "v_proj", "o_proj", "q_norm", "k_norm"):
                setattr(lay.self_attn, nm, _Holder(self._params[p + f"self_attn.{nm}.weight"]))
            lay.mlp = nn.Module()
            for nm in ("gate_proj", "up_proj", "down_proj"):
                setattr(lay.mlp, nm, _Holder(self._params[p + f"mlp.{nm}.weight"]))
            lay.input_layernorm = _Holder(self._params[p + "input_layernorm.weight"])
            lay.post_attention_layernorm = _Holder(self._params[p + "post_attention_layernorm.weight"])
            self.model.layers.append(lay)
        self.model.norm = _Holder(self._params["model.norm.weight"])
        self.lm_head = _Holder(self._params["model.embed_tokens.weight"] if d.tie_word_embeddings
                               else self._params["lm_head.weight"])
        self._anchor = torch.zeros((), device=dev, requires_grad=True)
        self._rope = {}
        self._cdims = _lib.Dims(d.vocab_size, h, I, d.num_hidden_layers, d.num_attention_heads, d.num_key_value_heads,
                                d.head_dim, int(d.tie_word_embeddings), d.rms_norm_eps, 0)
        self._cparams, self._clayers = self._c_struct(self.flat)
        self._cgrads = None
        self._stage_cb = None  # python callable(stage) set by the data-parallel wrapper
        self.overlap_dw = True
        self._side_stream = None
        # The attention kernels take a valid-prefix length per sequence, i.e. RIGHT padding -- what the collator emits
        # (data.py:280-327).  ``forward`` checks the mask for that (one host read); a caller that has already checked
        # (DistillationTrainer folds it into the row-count read it needs anyway) passes ``padding_checked=True``.
        self.validate_padding = True
        # A FROZEN model (the teacher: train.py:165-169) runs its inference forward with every decoder layer's RMSNorm
        # gains folded into the q|k|v / gate|up weights (SD_SAVE_NONE_FOLDED: 1 instead of 2L+1 norm launches per pass);
        # the folded copies are rebuilt whenever the flat parameter buffer has been written to (version counter).
        self.fold_norm_gains = True
        self._folded = None  # (flat._version, folded weight buffer, Params, Layers)
        self._lora = None    # lora.LoraState once lora.get_lora_model() has attached an adapter (train.py:180-202)
        if init_std:
            self.init_weights(seed, init_std)

    # ------------------------------------------------------------------------------- construction
    def _apply(self, fn, recurse=True):
        """``.to(device)`` / ``.cuda()`` (HF Trainer moves the model to ``args.device``): move the FLAT buffers and
        re-point the HF-named parameters at them, so they stay views of one buffer.  Other dtypes are refused."""
        new_flat = fn(self.flat)
        if new_flat.dtype != torch.bfloat16:
            raise TypeError("HipQwen3ForCausalLM holds bf16 parameters (train.py:174 loads the student in bf16)")
        if new_flat.device == self.flat.device:
            return self
        self.flat = new_flat
        for name, (o, n, shape) in self._slices.items():
            self._params[name].data = new_flat[o:o + n].view(shape)
        if self.flat_grad is not None:
            self.flat_grad = fn(self.flat_grad)
            self._cgrads, self._cglayers = self._c_struct(self.flat_grad)
            if self._grads_live:
                for name, (o, n, shape) in self._slices.items():
                    self._params[name].grad = self.flat_grad[o:o + n].view(shape)
        self._anchor = torch.zeros((), device=new_flat.device, requires_grad=True)
        self._cparams, self._clayers = self._c_struct(self.flat)
        self._rope, self._side_stream = {}, None
        if self._lora is not None:
            self._lora.rebind(fn)
        return self

    @staticmethod
    def _hf_config(d):
        try:
            from transformers import Qwen3Config
            return Qwen3Config(vocab_size=d.vocab_size, hidden_size=d.hidden_size, intermediate_size=d.intermediate_size,
                               num_hidden_layers=d.num_hidden_layers, num_attention_heads=d.num_attention_heads,
                               num_key_value_heads=d.num_key_value_heads, head_dim=d.head_dim,
                               rms_norm_eps=d.rms_norm_eps, rope_theta=d.rope_theta,
                               tie_word_embeddings=d.tie_word_embeddings, attention_bias=False)
        except Exception:  # transformers absent / different signature: the config is informational only
            return d

    def _c_struct(self, flat):
        d = self.dims
        base = flat.data_ptr()

        def ptr(name):
            return base + self._slices[name][0] * 2
        layers = (_lib.Layer * d.num_hidden_layers)()
        for l in range(d.num_hidden_layers):
            p = f"model.layers.{l}."
            layers[l].wqkv = ptr(p + "self_attn.q_proj.weight")
            layers[l].wo = ptr(p + "self_attn.o_proj.weight")
            layers[l].wgu = ptr(p + "mlp.gate_proj.weight")
            layers[l].wdown = ptr(p + "mlp.down_proj.weight")
            layers[l].q_gain = ptr(p + "self_attn.q_norm.weight")
            layers[l].k_gain = ptr(p + "self_attn.k_norm.weight")
            layers[l].ln1 = ptr(p + "input_layernorm.weight")
            layers[l].ln2 = ptr(p + "post_attention_layernorm.weight")
        params = _lib.Params()
        params.embed = ptr("model.embed_tokens.weight")
        params.lm_head = params.embed if d.tie_word_embeddings else ptr("lm_head.weight")
        params.final_norm = ptr("model.norm.weight")
        params.layers_host = C.cast(layers, C.POINTER(_lib.Layer))
        return params, layers

    @torch.no_grad()
    def init_weights(self, seed=0, std=0.02):
        """HF default init: N(0, std) matrices, unit norm gains; one CPU generator -> same on every rank."""
        g = torch.Generator().manual_seed(seed)
        for name, p in self._params.items():
            if p.dim() == 1:
                p.fill_(1.0)
            else:
                chunk = 1 << 24
                flat = p.view(-1)
                for s in range(0, flat.numel(), chunk):
                    n = min(chunk, flat.numel() - s)
                    flat[s:s + n].copy_((torch.randn(n, generator=g) * std).to(torch.bfloat16))

    @torch.no_grad()
    def load_hf_state_dict(self, sd):
        """Copy tensors from a dict with HF key names (fp32 or bf16, any device)."""
        for name, p in self._params.items():
            if name == "lm_head.weight" and name not in sd:
                continue
            p.copy_(sd[name].to(torch.bfloat16))

    # ------------------------------------------------------------------------------ checkpointing
    # The reference checkpoints through HF Trainer (save_strategy="epoch", load_best_model_at_end=True,
    # save_total_limit=3: train.py:341-345), i.e. ``save_pretrained`` of an HF Qwen3ForCausalLM: a directory with
    # ``config.json`` + ``model.safetensors`` whose keys are the HF names and which does NOT hold ``lm_head.weight``
    # when the head is tied (HF drops ``_tied_weights_keys``).  The same directory is written and read here.
    _tied_weights_keys = {"lm_head.weight": "model.embed_tokens.weight"}

    def _tied_head_key(self, prefix=""):
        return prefix + "lm_head.weight" if self.dims.tie_word_embeddings else None

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        """HF key names -> tensors (views of the flat buffer).  A tied ``lm_head.weight`` is left out, as in an HF
        checkpoint: it is the same memory as ``model.embed_tokens.weight`` and safetensors refuses aliases."""
        if self._lora is not None:   # a LoRA student checkpoints its adapter (peft's adapter state dict), lora.py
            return self._lora.state_dict(prefix)
        sd = super().state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        tied = self._tied_head_key(prefix)
        if tied is not None:
            sd.pop(tied, None)
        return sd

    def load_state_dict(self, state_dict, strict=True, assign=False):
        """Copy HF-named tensors (any float dtype / device) into the flat buffer.  Returns torch's
        ``_IncompatibleKeys``; a tied ``lm_head.weight`` is neither required nor, when present, unexpected
        (it must then equal the embedding, which is what gets loaded)."""
        from torch.nn.modules.module import _IncompatibleKeys
        if self._lora is not None:
            return self._lora.load_state_dict(state_dict, strict)
        tied = self._tied_head_key()
        want = [k for k in self._params if k != tied] if tied else list(self._params)
        if tied and "lm_head.weight" in self._params:
            want = [k for k in want if k != "lm_head.weight"]
        missing = [k for k in want if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._params and k != tied]
        bad = [k for k in want if k in state_dict and tuple(state_dict[k].shape) != tuple(self._params[k].shape)]
        if bad:
            raise RuntimeError("size mismatch for " + ", ".join(
                f"{k}: checkpoint {tuple(state_dict[k].shape)} vs model {tuple(self._params[k].shape)}" for k in bad))
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for {type(self).__name__}: missing {missing}, "
                               f"unexpected {unexpected}")
        with torch.no_grad():
            for k in want:
                if k in state_dict:
                    self._params[k].copy_(state_dict[k])
        return _IncompatibleKeys(missing, unexpected)

    def save_pretrained(self, save_directory, state_dict=None, safe_serialization=True, **kwargs):
        """HF-loadable checkpoint directory: ``config.json`` (+ ``model.safetensors`` / ``pytorch_model.bin``).
        ``AutoModelForCausalLM.from_pretrained(dir)`` and ``HipQwen3ForCausalLM.from_pretrained(dir)`` both read it."""
        if self._lora is not None:   # adapter_config.json + adapter_model.safetensors, as PeftModel.save_pretrained
            return self._lora.save_pretrained(save_directory, state_dict, safe_serialization)
        os.makedirs(save_directory, exist_ok=True)
        sd = self.state_dict() if state_dict is None else dict(state_dict)
        tied = self._tied_head_key()
        if tied:
            sd.pop(tied, None)
        sd = {k: v.detach().contiguous() for k, v in sd.items()}
        if hasattr(self.config, "save_pretrained"):
            if getattr(self.config, "architectures", None) is None:
                self.config.architectures = ["Qwen3ForCausalLM"]
            self.config.save_pretrained(save_directory)
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(save_directory, "model.safetensors"), metadata={"format": "pt"})
        else:
            torch.save(sd, os.path.join(save_directory, "pytorch_model.bin"))

    @classmethod
    def from_pretrained(cls, directory, device="cuda", **kwargs):
        """Build from an HF checkpoint directory (what ``save_pretrained`` above or HF itself wrote); local only."""
        import json
        from transformers import AutoConfig
        c = AutoConfig.from_pretrained(directory)
        dims = Qwen3Dims.from_hf_config(c)
        m = cls(dims, device=device, config=c, init_std=0)
        st = os.path.join(directory, "model.safetensors")
        idx = os.path.join(directory, "model.safetensors.index.json")
        if os.path.isfile(st):
            from safetensors.torch import load_file
            m.load_state_dict(load_file(st, device="cpu"))
        elif os.path.isfile(idx):
            from safetensors.torch import load_file
            files = sorted(set(json.load(open(idx))["weight_map"].values()))
            sd = {}
            for f in files:
                sd.update(load_file(os.path.join(directory, f), device="cpu"))
            m.load_state_dict(sd)
        else:
            m.load_state_dict(torch.load(os.path.join(directory, "pytorch_model.bin"), map_location="cpu",
                                         weights_only=True))
        return m

    # ------------------------------------------------------------------ HF/Trainer protocol no-ops
    def train(self, mode: bool = True):
        """``nn.Module.train`` walks every submodule; the ~400 HF-named ones here only hold parameter views, and HF's
        ``training_step`` calls ``model.train()`` on every micro-step (1.4 ms of host time with the GPU idle)."""
        self.training = mode
        return self

    def gradient_checkpointing_enable(self, gradient_checkpointing_kwargs=None, **_):
        """train.py:204-208 / TrainingArguments(gradient_checkpointing=True) (train.py:340), on by default in the reference.

        Checkpointing trades a second forward of every decoder layer for activation memory.  The runner implements it
        at the same granularity as HF (one decoder layer: only the residual stream entering each layer is kept,
        ``SD_SAVE_LAYER_INPUTS``), with bit-identical gradients, but WHEN it recomputes is a memory policy, chosen by
        ``gradient_checkpointing_kwargs={"recompute": ...}`` (or the environment variable ``SD_RECOMPUTE``):

        * ``"auto"`` (default): recompute only when keeping every activation of this batch would take more than
          ``recompute_fraction`` (default 0.25) of the device's memory -- never at the reference's shapes on 288 GB
          (2.6 GB at B=4,T=512; 10.4 GB at B=4,T=2048), so the default path pays nothing for the flag;
        * ``"always"``: what the flag means on a 24-80 GB card (+1 student forward per step, activations / ~9);
        * ``"never"``.
        """
        kw = dict(gradient_checkpointing_kwargs or {})
        # a later call without the key (HF Trainer calls this again with its own kwargs) keeps an earlier choice
        policy = kw.get("recompute") or os.environ.get("SD_RECOMPUTE") or self.recompute_policy
        if policy not in ("auto", "always", "never"):
            raise ValueError(f"recompute policy {policy!r}: expected 'auto', 'always' or 'never'")
        self.recompute_policy = policy
        self.recompute_fraction = float(kw.get("recompute_fraction", self.recompute_fraction))
        self.gradient_checkpointing = True

    def gradient_checkpointing_disable(self):
        self.gradient_checkpointing = False

    @property
    def is_gradient_checkpointing(self):
        return self.gradient_checkpointing

    def _wants_recompute(self, B, T, device):
        if not self.gradient_checkpointing or self.recompute_policy == "never":
            return False
        if self.recompute_policy == "always":
            return True
        full = load_lib().sd_qwen3_acts_bytes(C.byref(self._cdims), B, T, SAVE_ALL)
        return full > self.recompute_fraction * torch.cuda.get_device_properties(device).total_memory

    def enable_input_require_grads(self):
        pass

    def get_input_embeddings(self):
        return self.model.embed_tokens

    # --------------------------------------------------------------------------------- execution
    def _tables(self, T, device):
        key = (T, str(device))
        if key not in self._rope:
            self._rope[key] = rope_tables(T, device, self.dims.rope_theta)
        return self._rope[key]

    @torch.no_grad()
    def _folded_params(self):
        """C parameter struct whose wqkv / wgu point at W diag(g) (g = the RMSNorm gain in front of the projection,
        HF:59-64, 252-254, 81-83), or None when this model must not or cannot fold."""
        if not self.fold_norm_gains or self._lora is not None or any(p.requires_grad for p in self._params.values()):
            return None
        if not load_lib().sd_qwen3_fold_supported(C.byref(self._cdims)):
            return None
        ver = self.flat._version
        if self._folded is not None and self._folded[0] == ver and self._folded[1].device == self.flat.device:
            return self._folded[2]
        d = self.dims
        h, I = d.hidden_size, d.intermediate_size
        nq, ng = (d.q_dim + 2 * d.kv_dim) * h, 2 * I * h
        buf = torch.empty(d.num_hidden_layers * (nq + ng), dtype=torch.bfloat16, device=self.flat.device)
        params, layers = self._c_struct(self.flat)
        for l in range(d.num_hidden_layers):
            p = f"model.layers.{l}."
            o, _, _ = self._slices[p + "self_attn.q_proj.weight"]
            wqkv = self.flat[o:o + nq].view(-1, h)       # q | k | v rows are contiguous in the flat layout
            o, _, _ = self._slices[p + "mlp.gate_proj.weight"]
            wgu = self.flat[o:o + ng].view(-1, h)        # gate | up likewise
            base = l * (nq + ng)
            fq, fg = buf[base:base + nq].view(-1, h), buf[base + nq:base + nq + ng].view(-1, h)
            fq.copy_(wqkv.float() * self._params[p + "input_layernorm.weight"].float()[None, :])
            fg.copy_(wgu.float() * self._params[p + "post_attention_layernorm.weight"].float()[None, :])
            layers[l].wqkv = buf.data_ptr() + base * 2
            layers[l].wgu = buf.data_ptr() + (base + nq) * 2
        self._folded = (ver, buf, params, layers)
        return params

    def _run_forward(self, input_ids, kv_len, save, rows=None, concurrent=False):
        lib = load_lib()
        B, T = input_ids.shape
        dev = input_ids.device
        cos, sin = self._tables(T, dev)
        cparams = self._cparams
        if self._lora is not None:
            self._lora.ensure_merged()   # W_eff = W_res + s B A, rebuilt only after A / B have changed
        if save == SAVE_NONE:
            folded = self._folded_params()
            if folded is not None:
                cparams, save = folded, SAVE_NONE_FOLDED
        nbytes = lib.sd_qwen3_acts_bytes(C.byref(self._cdims), B, T, int(save))
        acts = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        if rows is None:
            logits = torch.empty(B, T, self.dims.vocab_size, dtype=torch.bfloat16, device=dev)
        else:
            logits = torch.empty(rows.numel(), self.dims.vocab_size, dtype=torch.bfloat16, device=dev)
        check(lib.sd_qwen3_forward_rows(C.byref(self._cdims), C.byref(cparams), input_ids.data_ptr(), _p(kv_len),
                                        cos.data_ptr(), sin.data_ptr(), acts.data_ptr(), nbytes, logits.data_ptr(),
                                        _p(rows), 0 if rows is None else rows.numel(), B, T,
                                        int(save) | (FWD_CONCURRENT if concurrent else 0), _stream()),
              "sd_qwen3_forward_rows")
        return logits, acts

    def _ensure_grads(self):
        """Attach HF-named .grad views of the flat gradient buffer; returns accumulate flag."""
        if self.flat_grad is None:
            self.flat_grad = torch.zeros_like(self.flat)
            self._cgrads, self._cglayers = self._c_struct(self.flat_grad)
        first = next((p for p in self._params.values() if p.requires_grad), None)
        # (optimizer.zero_grad(set_to_none=True) clears .grad without going through zero_grad() below)
        accumulate = self._grads_live and (first is None or first.grad is not None)
        if not accumulate:
            for name, (o, n, shape) in self._slices.items():
                if self._params[name].requires_grad:  # (a LoRA student's frozen base: the buffer still receives dW)
                    self._params[name].grad = self.flat_grad[o:o + n].view(shape)
            self._grads_live = True
        return accumulate

    def _run_backward(self, input_ids, kv_len, acts, dlogits, rows=None, recompute=False):
        lib = load_lib()
        B, T = input_ids.shape
        accumulate = self._ensure_grads()
        red = getattr(self, "_reducer", None)
        dx0 = None
        if red is not None:
            red.begin_step()
            if red.wants_split_embedding():
                # tied embedding/lm_head gradient: the dense lm_head part is all-reduced at the START of
                # backward; only the B*T rows touched by the embedding lookup are exchanged at the end
                dx0 = torch.empty(B * T, self.dims.hidden_size, dtype=torch.bfloat16, device=input_ids.device)
                eo, en, eshape = self._slices["model.embed_tokens.weight"]
                red.set_embedding_exchange(input_ids.reshape(-1), dx0, self.flat_grad[eo:eo + en].view(eshape))
        if not dlogits.is_contiguous():
            dlogits = dlogits.contiguous()
        cos, sin = self._tables(T, input_ids.device)
        sbytes = lib.sd_qwen3_bwd_scratch_bytes(C.byref(self._cdims), B, T)
        scratch = torch.empty(sbytes, dtype=torch.uint8, device=input_ids.device)
        user_cb = self._stage_cb
        cb = _lib.STAGE_CB((lambda stage, _u: user_cb(stage)) if user_cb else 0)
        check(lib.sd_qwen3_backward_rows(C.byref(self._cdims), C.byref(self._cparams), C.byref(self._cgrads),
                                         input_ids.data_ptr(), _p(kv_len), cos.data_ptr(), sin.data_ptr(),
                                         acts.data_ptr(), acts.numel(), dlogits.data_ptr(), _p(rows),
                                         0 if rows is None else rows.numel(), scratch.data_ptr(), sbytes, B, T,
                                         (BWD_ACCUMULATE if accumulate else 0) | (BWD_RECOMPUTE if recompute else 0),
                                         _p(dx0), cb, None, self._side_stream_ptr(input_ids.device),
                                         _stream()),
              "sd_qwen3_backward_rows")
        if red is not None:
            red.finish()
        if self._lora is not None:
            self._lora.grads_stale = True

    def finalize_grads(self):
        """Called by the optimizer / the clipping hook before they read gradients: a LoRA student projects the
        accumulated weight gradient onto its adapter (dA = s B^T dW, dB = s dW A^T) here, once per optimizer step."""
        if self._lora is not None:
            self._lora.project_grads()

    def optim_segments(self, split_decay=False):
        """What the fused optimizer updates: a list of (kind, params, grads, decays, extra) over flat buffers, in a
        fixed order.  kind "bf16": bf16 parameters / moments (sd_adamw_bf16); "f32_shadow": fp32 masters with bf16
        gradients and shadows (sd_adamw_f32_shadow; extra = (shadow, shadow_scaled, scale))."""
        g = self.flat_grad
        if self._lora is not None:
            return self._lora.optim_segments()
        if not split_decay:
            return [("bf16", self.flat, g, True, None)]
        return [("bf16", self.flat[a:b], None if g is None else g[a:b], m, None) for a, b, m in self._decay_runs()]

    def _decay_runs(self):
        """Contiguous runs of matrices / gains in the flat layout (weight decay applies to matrices only)."""
        runs, cur = [], None
        for name, (o, n, shape) in self._slices.items():
            is_mat = len(shape) == 2
            n8 = (n + 7) // 8 * 8
            if cur is not None and cur[2] == is_mat and cur[1] == o:
                cur[1] = o + n8
            else:
                cur = [o, o + n8, is_mat]
                runs.append(cur)
        return [(a, b, m) for a, b, m in runs]

    def _side_stream_ptr(self, device):
        """Second HIP stream for the weight-gradient GEMMs (they overlap the dX chain); None disables it."""
        if not self.overlap_dw:
            return None
        if self._side_stream is None:
            from . import ops
            self._side_stream = ops.concurrent_stream(device, "dw")
        return self._side_stream.cuda_stream

    def forward(self, input_ids=None, attention_mask=None, labels=None, logit_rows=None, **kwargs):
        """Returns an object with ``.logits`` [B,T,V] (bf16).  ``labels`` is accepted and ignored: the
        reference leaves it in ``inputs`` at train.py:54, which only makes HF compute an unused CE.
        ``logit_rows`` (int64 [R], flat b*T+t indices, unique): apply the lm_head to those rows only and return
        ``.logits`` [R,V] -- the training step passes the rows the loss reads (``ops.loss_rows``).
        ``concurrent=True`` (keyword): the caller runs another pass beside this one on a second stream (the frozen
        teacher beside the student, as DistillationTrainer does): SD_FWD_CONCURRENT, launches sized for a shared GPU."""
        ids = _need(input_ids.to(torch.int64), torch.int64, "input_ids")
        rows = None
        if logit_rows is not None:
            rows = _need(logit_rows.to(device=ids.device, dtype=torch.int64), torch.int64, "logit_rows")
            if rows.dim() != 1 or rows.numel() == 0 or rows.numel() > ids.numel():
                raise ValueError("logit_rows must be a non-empty 1-D tensor of at most B*T row indices")
        kv_len = None
        if attention_mask is not None:
            am = attention_mask.to(ids.device)
            if am.shape != ids.shape:
                raise ValueError(f"attention_mask {tuple(am.shape)} != input_ids {tuple(ids.shape)}")
            if self.validate_padding and not kwargs.get("padding_checked", False) and bool(left_padded(am)):
                raise ValueError("attention_mask is not right-padded (a 1 follows a 0): the HIP attention kernels take a "
                                 "valid-prefix length per sequence, as ProcessedDataCollator produces (data.py:280-327)")
            kv_len = am.sum(-1).to(torch.int32).contiguous()
        if torch.is_grad_enabled() and (self._lora is not None or any(p.requires_grad for p in self._params.values())):
            logits = _DecoderFn.apply(self._anchor, ids, kv_len, self, rows, bool(kwargs.get("concurrent", False)))
        else:
            logits, _ = self._run_forward(ids, kv_len, save=SAVE_NONE, rows=rows,
                                          concurrent=bool(kwargs.get("concurrent", False)))
        return CausalLMOutput(logits=logits)

    def zero_grad(self, set_to_none: bool = True):
        # keep the flat buffer; the next backward overwrites (accumulate=0) instead of adding
        self._grads_live = False
        for p in self._params.values():
            p.grad = None
