"""LoRA student (train.py:180-202, flags train.py:470-487): the counterpart of ``peft.get_peft_model(student, LoraConfig(
r, lora_alpha, target_modules = the seven projections, modules_to_save = [embed_tokens, lm_head], lora_dropout = 0,
bias = "none", use_rslora, init_lora_weights))`` for the flat-buffer HIP student.

PARITY UNPINNED: ``peft`` is an unvendored, unpinned dependency of the reference (requirements.txt:15) and is not in the
build image; ``oracle/lora.py`` restates its published LoRA layer and the GPU tests compare against that.

Same function, different evaluation (sd_lora.hip): the adapter is applied to the MERGED weight

    W_eff = W_res + s B A      s = lora_alpha / sqrt(r)  (rsLoRA)  or  lora_alpha / r

rebuilt once per optimizer step, the decoder runs its ordinary kernels on W_eff, the backward writes the ordinary full
weight gradient dW into the flat gradient buffer (accumulating over micro-batches, all-reduced by the data-parallel
wrapper as always), and before the optimizer runs dA = s B^T dW, dB = s dW A^T.  On MI355X the dW GEMMs run at ~1 PFLOP/s
while peft's form adds six rank-32 GEMMs per projection per micro-batch (1176 skinny launches); merge and projection are
three HBM-bound launches per OPTIMIZER step.

What trains, as under peft: A [r, in] and B [out, r] of every target (fp32 masters with fp32 AdamW moments -- peft keeps
adapter weights in fp32 next to a bf16 base, ``autocast_adapter_dtype``), plus full copies of ``embed_tokens`` and
``lm_head`` (bf16, bf16 moments like the rest of the bf16 model, quirk Q5).  peft copies the two modules independently,
so a TIED student trains them UNTIED: ``get_lora_model`` rebuilds a tied model with a separate ``lm_head.weight``.
Everything else is frozen.  The merged weight is stored in bf16 like every other weight of the bf16 model (deviation:
peft adds the adapter's output to the base output in bf16 activations instead).

Checkpoints: ``save_pretrained`` writes peft's adapter layout (``adapter_config.json`` + ``adapter_model.safetensors``
with keys ``base_model.model.<module>.lora_A.weight`` / ``lora_B.weight`` and the two saved modules).  With a PiSSA
init the adapter is relative to the RESIDUAL base, exactly as under peft; ``merge_and_unload()`` gives the plain model.
"""
from __future__ import annotations

import copy
import ctypes as C
import json
import math
import os
from dataclasses import asdict, dataclass, replace
from typing import Sequence, Union

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, load_lib
from .ops import _stream
from .qwen3 import HipQwen3ForCausalLM, _Holder

TARGETS = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")   # train.py:185-193
SAVED = ("embed_tokens", "lm_head")                                                        # train.py:194
_SAVED_KEY = {"embed_tokens": "model.embed_tokens.weight", "lm_head": "lm_head.weight"}
_PEFT_PREFIX = "base_model.model."


@dataclass
class LoraConfig:
    """The fields of ``peft.LoraConfig`` the reference sets (train.py:182-200), with the reference's defaults
    (train.py:474-487: r 32, alpha 64, rsLoRA on, PiSSA init)."""
    r: int = 32
    lora_alpha: int = 64
    target_modules: Sequence[str] = TARGETS
    modules_to_save: Sequence[str] = SAVED
    lora_dropout: float = 0.0
    bias: str = "none"
    task_type: str = "CAUSAL_LM"
    use_rslora: bool = True
    init_lora_weights: Union[bool, str] = "pissa"

    @property
    def scaling(self) -> float:
        return self.lora_alpha / math.sqrt(self.r) if self.use_rslora else self.lora_alpha / self.r

    def validate(self):
        if self.r <= 0 or self.r > 128:
            raise ValueError("LoRA rank must be in 1..128 (sd_lora.hip holds the whole rank in one register tile)")
        if self.lora_dropout != 0.0:
            raise NotImplementedError("lora_dropout != 0 (the reference passes 0, train.py:195): the merged-weight form "
                                      "has no place for a dropout between x and A")
        if self.bias != "none":
            raise NotImplementedError('bias != "none" (the reference passes "none", train.py:196; Qwen3 has no biases)')
        bad = [t for t in self.target_modules if t not in TARGETS]
        if bad or not self.target_modules:
            raise ValueError(f"target_modules {bad or '[]'}: expected a non-empty subset of {TARGETS}")
        bad = [t for t in self.modules_to_save if t not in SAVED]
        if bad:
            raise ValueError(f"modules_to_save {bad}: expected a subset of {SAVED}")
        m = self.init_lora_weights
        if not (m in (True, "default", "gaussian", "pissa") or (isinstance(m, str) and m.startswith("pissa_niter_"))):
            raise ValueError(f"init_lora_weights={m!r}: expected True/'default', 'gaussian', 'pissa' or 'pissa_niter_N'")


def _init_pair(W, r, scale, method, gen):
    """peft ``reset_lora_parameters`` / ``pissa_init`` for one weight W [out, in] (fp32, on its device): (A, B, W_res)."""
    out_f, in_f = W.shape
    dev = W.device
    if method in (True, "default"):    # kaiming_uniform_(a = sqrt 5) on [r, in]: U(-1/sqrt(in), 1/sqrt(in))
        A = ((torch.rand(r, in_f, generator=gen) * 2 - 1) / math.sqrt(in_f)).to(dev)
        return A, torch.zeros(out_f, r, device=dev), None
    if method == "gaussian":
        return (torch.randn(r, in_f, generator=gen) / r).to(dev), torch.zeros(out_f, r, device=dev), None
    if method == "pissa":
        U, S, Vh = torch.linalg.svd(W, full_matrices=False)
        Ur, Sr, Vhr = U[:, :r], S[:r] / scale, Vh[:r]
    else:
        Ur, Sr, Vr = torch.svd_lowrank(W, r, niter=int(method.split("_niter_")[-1]))
        Sr, Vhr = Sr / scale, Vr.t()
    A = torch.sqrt(Sr)[:, None] * Vhr
    B = Ur * torch.sqrt(Sr)[None, :]
    return A, B, W - scale * (B @ A)


class LoraState:
    """Buffers, launch plan and bookkeeping of one attached adapter (``model._lora``)."""

    def __init__(self, model: HipQwen3ForCausalLM, cfg: LoraConfig, seed: int = 0):
        cfg.validate()
        if not model.flat.is_cuda:
            raise RuntimeError("speech_distill_amd: the LoRA student lives on the GPU (no CPU fallback)")
        self.model, self.cfg = model, cfg
        self.r, self.scale = int(cfg.r), float(cfg.scaling)
        self.r_pad = 32 if cfg.r <= 32 else (64 if cfg.r <= 64 else 128)
        d = model.dims
        self.targets = []          # (HF weight name, out, in)
        for l in range(d.num_hidden_layers):
            for t in TARGETS:
                if t in cfg.target_modules:
                    grp = "mlp" if t in ("gate_proj", "up_proj", "down_proj") else "self_attn"
                    name = f"model.layers.{l}.{grp}.{t}.weight"
                    out_f, in_f = model._slices[name][2]
                    if in_f % 128 or out_f % 32:
                        raise _lib.SdHipError(f"{name} [{out_f}, {in_f}]: sd_lora.hip needs in % 128 == 0 and out % 32 == 0")
                    self.targets.append((name, out_f, in_f))
        self.saved = [_SAVED_KEY[m] for m in SAVED if m in cfg.modules_to_save]
        dev = model.flat.device
        rp = self.r_pad
        self.a_off, self.b_off, self.w_off = [], [], []
        na = nb = nw = 0
        for _, out_f, in_f in self.targets:
            self.a_off.append(na), self.b_off.append(nb), self.w_off.append(nw)
            na, nb, nw = na + rp * in_f, nb + out_f * rp, nw + out_f * in_f
        self.n_a, self.n_b = na, nb
        self.b_off = [na + o for o in self.b_off]
        self.master = torch.zeros(na + nb, dtype=torch.float32, device=dev)      # A_0 .. A_n | B_0 .. B_n (padded to r_pad)
        self.shadow = torch.zeros(na + nb, dtype=torch.bfloat16, device=dev)     # bf16(p)
        self.shadow_scaled = torch.zeros_like(self.shadow)                       # bf16(s p)
        self.grad = torch.zeros_like(self.shadow)                                # dA_0 .. | dB_0 ..
        self.base = torch.empty(nw, dtype=torch.bfloat16, device=dev)            # W_res of every target
        if model.flat_grad is None:     # the plan holds pointers into the gradient buffer
            model.flat_grad = torch.zeros_like(model.flat)
            model._cgrads, model._cglayers = model._c_struct(model.flat_grad)
        self._init_weights(seed)
        self._register_parameters()
        self._build_plan()
        self.refresh_shadows()
        self.grads_stale = False

    # ------------------------------------------------------------------------------------------------ set-up
    def a_view(self, i, buf=None):
        _, _, in_f = self.targets[i]
        return (self.master if buf is None else buf)[self.a_off[i]:self.a_off[i] + self.r_pad * in_f].view(self.r_pad, in_f)

    def b_view(self, i, buf=None):
        _, out_f, _ = self.targets[i]
        return (self.master if buf is None else buf)[self.b_off[i]:self.b_off[i] + out_f * self.r_pad].view(out_f, self.r_pad)

    def base_view(self, i):
        _, out_f, in_f = self.targets[i]
        return self.base[self.w_off[i]:self.w_off[i] + out_f * in_f].view(out_f, in_f)

    @torch.no_grad()
    def _init_weights(self, seed):
        gen = torch.Generator().manual_seed(seed)   # CPU generator: the same adapter on every rank
        method = self.cfg.init_lora_weights
        for i, (name, out_f, in_f) in enumerate(self.targets):
            W = self.model._params[name].data
            A, B, res = _init_pair(W.float(), self.r, self.scale, method, gen)
            self.a_view(i)[:self.r].copy_(A)
            self.b_view(i)[:, :self.r].copy_(B)
            self.base_view(i).copy_(W if res is None else res)   # PiSSA: the residual, rounded to the base dtype

    def _register_parameters(self):
        m = self.model
        for p in m._params.values():
            p.requires_grad_(False)
        for key in self.saved:
            m._params[key].requires_grad_(True)
        self.params = {}           # peft-style key (without prefix) -> Parameter
        for i, (name, out_f, in_f) in enumerate(self.targets):
            l, grp, t = name.split(".")[2:5]
            holder = getattr(getattr(m.model.layers[int(l)], grp), t)
            A = nn.Parameter(self.a_view(i)[:self.r])
            B = nn.Parameter(self.b_view(i)[:, :self.r])
            holder.lora_A, holder.lora_B = _Holder(A), _Holder(B)
            stem = name[:-len("weight")]
            self.params[stem + "lora_A.weight"], self.params[stem + "lora_B.weight"] = A, B

    def _build_plan(self):
        lib = load_lib()
        m = self.model
        n = len(self.targets)
        arr = (_lib.LoraTarget * n)()
        fb, gb = m.flat.data_ptr(), m.flat_grad.data_ptr()
        for i, (name, out_f, in_f) in enumerate(self.targets):
            o = m._slices[name][0]
            t = arr[i]
            t.w_res = self.base.data_ptr() + self.w_off[i] * 2
            t.w_out, t.w_grad = fb + o * 2, gb + o * 2
            t.a_shadow = self.shadow.data_ptr() + self.a_off[i] * 2
            t.a_scaled = self.shadow_scaled.data_ptr() + self.a_off[i] * 2
            t.b_scaled = self.shadow_scaled.data_ptr() + self.b_off[i] * 2
            t.d_a = self.grad.data_ptr() + self.a_off[i] * 2
            t.d_b = self.grad.data_ptr() + self.b_off[i] * 2
            t.out_features, t.in_features = out_f, in_f
        nbytes = lib.sd_lora_plan_bytes(n)
        self._plan_host = C.create_string_buffer(nbytes)
        check(lib.sd_lora_plan_build(arr, n, self.r_pad, self._plan_host, nbytes), "sd_lora_plan_build")
        self._plan_dev = torch.frombuffer(bytearray(self._plan_host.raw), dtype=torch.uint8).to(m.flat.device)
        self._plan_for = (fb, gb)

    def rebind(self, fn):
        """The model's flat buffers have moved (``.to(device)``): move the adapter's buffers along and re-plan."""
        for k in ("master", "shadow", "shadow_scaled", "grad", "base"):
            setattr(self, k, fn(getattr(self, k)))
        for i, (name, _, _) in enumerate(self.targets):
            stem = name[:-len("weight")]
            self.params[stem + "lora_A.weight"].data = self.a_view(i)[:self.r]
            self.params[stem + "lora_B.weight"].data = self.b_view(i)[:, :self.r]
        self._build_plan()
        self._shadow_version = None
        self.dirty = True

    # ----------------------------------------------------------------------------------------------- running
    @torch.no_grad()
    def refresh_shadows(self):
        """bf16 operands of the kernels from the fp32 masters (after init / a load / a write through the Parameters;
        the optimizer kernel writes them itself)."""
        self.shadow.copy_(self.master)
        self.shadow_scaled.copy_(self.master * self.scale)
        self._shadow_version = self.master._version
        self.dirty = True

    def mark_updated(self):
        """The optimizer kernel has rewritten masters and shadows."""
        self.dirty = True

    def ensure_merged(self):
        if self.master._version != self._shadow_version:
            self.refresh_shadows()
        if self._plan_for != (self.model.flat.data_ptr(), self.model.flat_grad.data_ptr()):
            self._build_plan()
        if self.dirty:
            check(load_lib().sd_lora_merge(self._plan_dev.data_ptr(), self._plan_host, _stream()), "sd_lora_merge")
            self.dirty = False

    def project_grads(self):
        if not self.grads_stale:
            return
        if self.master._version != self._shadow_version:   # (A / B written between the backward and here)
            self.refresh_shadows()
        check(load_lib().sd_lora_project(self._plan_dev.data_ptr(), self._plan_host, _stream()), "sd_lora_project")
        self.grads_stale = False

    def optim_segments(self):
        m = self.model
        segs = []
        for key in self.saved:
            o, n, _ = m._slices[key]
            segs.append(("bf16", m.flat[o:o + n], m.flat_grad[o:o + n], True, None))
        segs.append(("f32_shadow", self.master, self.grad, True, (self.shadow, self.shadow_scaled, self.scale, self)))
        return segs

    def grads(self):
        """name -> gradient (views; call after ``model.finalize_grads()``): the adapter's dA [r, in] / dB [out, r] and
        the saved modules' gradients, under the names ``state_dict`` uses (without the peft prefix)."""
        out = {}
        for i, (name, _, _) in enumerate(self.targets):
            stem = name[:-len("weight")]
            out[stem + "lora_A.weight"] = self.a_view(i, self.grad)[:self.r]
            out[stem + "lora_B.weight"] = self.b_view(i, self.grad)[:, :self.r]
        for key in self.saved:
            o, n, shape = self.model._slices[key]
            out[key] = self.model.flat_grad[o:o + n].view(shape)
        return out

    # ------------------------------------------------------------------------------------------ checkpointing
    def state_dict(self, prefix=""):
        """peft's adapter state dict (``get_peft_model_state_dict``): lora_A / lora_B of every target + the saved modules."""
        sd = {}
        for k, p in self.params.items():
            sd[prefix + _PEFT_PREFIX + k] = p.detach()
        for key in self.saved:
            sd[prefix + _PEFT_PREFIX + key] = self.model._params[key].detach()
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict, strict=True):
        from torch.nn.modules.module import _IncompatibleKeys
        want = self.state_dict()
        missing = [k for k in want if k not in state_dict]
        unexpected = [k for k in state_dict if k not in want]
        bad = [k for k in want if k in state_dict and tuple(state_dict[k].shape) != tuple(want[k].shape)]
        if bad:
            raise RuntimeError("size mismatch for " + ", ".join(bad))
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading the adapter state_dict: missing {missing}, unexpected {unexpected}")
        for k, dst in want.items():
            if k in state_dict:
                dst.copy_(state_dict[k])
        self.refresh_shadows()
        return _IncompatibleKeys(missing, unexpected)

    def adapter_config(self):
        c = asdict(self.cfg)
        c["target_modules"], c["modules_to_save"] = list(c["target_modules"]), list(c["modules_to_save"])
        c.update(peft_type="LORA", fan_in_fan_out=False, inference_mode=True, base_model_name_or_path=None)
        return c

    def save_pretrained(self, save_directory, state_dict=None, safe_serialization=True):
        os.makedirs(save_directory, exist_ok=True)
        sd = self.state_dict() if state_dict is None else dict(state_dict)
        sd = {k: v.detach().contiguous() for k, v in sd.items()}
        with open(os.path.join(save_directory, "adapter_config.json"), "w") as f:
            json.dump(self.adapter_config(), f, indent=2, sort_keys=True)
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(save_directory, "adapter_model.safetensors"), metadata={"format": "pt"})
        else:
            torch.save(sd, os.path.join(save_directory, "adapter_model.bin"))

    def load_adapter(self, directory):
        st = os.path.join(directory, "adapter_model.safetensors")
        if os.path.isfile(st):
            from safetensors.torch import load_file
            return self.load_state_dict(load_file(st, device="cpu"))
        return self.load_state_dict(torch.load(os.path.join(directory, "adapter_model.bin"), map_location="cpu",
                                               weights_only=True))

    def trainable_parameters(self):
        """(trainable, total) as ``PeftModel.print_trainable_parameters`` counts them (train.py:203)."""
        m = self.model
        train = sum(p.numel() for p in self.params.values()) + sum(m._params[k].numel() for k in self.saved)
        total = sum(p.numel() for p in m._params.values()) + sum(p.numel() for p in self.params.values())
        return train, total


def _untied_copy(model: HipQwen3ForCausalLM) -> HipQwen3ForCausalLM:
    dims = replace(model.dims, tie_word_embeddings=False)
    cfg = copy.deepcopy(model.config)
    if hasattr(cfg, "tie_word_embeddings"):
        cfg.tie_word_embeddings = False
    new = HipQwen3ForCausalLM(dims, device=model.flat.device, config=cfg, init_std=0)
    with torch.no_grad():
        for name, p in model._params.items():
            new._params[name].copy_(p)
        new._params["lm_head.weight"].copy_(model._params["model.embed_tokens.weight"])
    for k in ("gradient_checkpointing", "recompute_policy", "recompute_fraction", "overlap_dw", "validate_padding",
              "training"):
        setattr(new, k, getattr(model, k))
    return new


def get_lora_model(model: HipQwen3ForCausalLM, config: LoraConfig, seed: int = 0) -> HipQwen3ForCausalLM:
    """``peft.get_peft_model`` for the HIP student (train.py:202).  Returns the model to train: ``model`` itself with the
    adapter attached, or -- when it ties ``lm_head`` to ``embed_tokens`` and both are in ``modules_to_save`` -- an untied
    copy (peft's ModulesToSaveWrapper deep-copies each module on its own, which unties them; drop ``model`` then)."""
    if not isinstance(model, HipQwen3ForCausalLM):
        raise TypeError("get_lora_model takes a HipQwen3ForCausalLM")
    if model._lora is not None:
        raise RuntimeError("this model already carries an adapter")
    config.validate()
    saved = set(config.modules_to_save)
    if model.dims.tie_word_embeddings:
        if saved == set(SAVED):
            model = _untied_copy(model)
        elif saved:
            raise NotImplementedError("modules_to_save with only one of embed_tokens / lm_head on a tied model")
    model._lora = LoraState(model, config, seed)
    return model


def print_trainable_parameters(model):
    train, total = model._lora.trainable_parameters()
    print(f"trainable params: {train:,d} || all params: {total:,d} || trainable%: {100 * train / total:.4f}")


@torch.no_grad()
def merge_and_unload(model: HipQwen3ForCausalLM) -> HipQwen3ForCausalLM:
    """``PeftModel.merge_and_unload``: the plain model holding W_res + s B A; every parameter trainable again."""
    st = model._lora
    if st is None:
        return model
    st.ensure_merged()
    for name, _, _ in st.targets:
        l, grp, t = name.split(".")[2:5]
        holder = getattr(getattr(model.model.layers[int(l)], grp), t)
        del holder.lora_A, holder.lora_B
    model._lora = None
    for p in model._params.values():
        p.requires_grad_(True)
    model.zero_grad()
    return model
