"""Fused AdamW over the model's flat bf16 parameter / gradient buffers.

The reference trains pure-bf16 parameters with the HF Trainer's default AdamW, i.e. bf16 moments
(train.py:174 loads the student in bf16, train.py:331-354 sets no ``optim``; quirk Q5), and clips the
global gradient norm to ``max_grad_norm`` (HF trainer.py:2539).  Here one ``sd_sumsq_bf16`` launch computes
the squared norm of the whole flat gradient and one ``sd_adamw_bf16`` launch updates all 604 M parameters
(fp32 arithmetic in registers, one rounding of p / m / v to bf16), with the clip coefficient read from device
memory -- no host synchronisation, ~8.5 GB of HBM traffic per step.

``DistillationTrainer.create_optimizer`` builds it in place of HF's default AdamW, and its ``_clip_grad_norm`` hook calls
``grad_norm(max_grad_norm)`` so that HF's clipping (one ``clip_grad_norm_`` over ~310 tensors, 10 ms of host time)
becomes one reduction plus a coefficient applied inside the update.  Stand-alone: ``FlatAdamW(student, clip=1.0)``.
Weight decay applies to matrices only (norm gains are excluded, as HF's parameter grouping does).

What it updates comes from ``model.optim_segments()``: for the fully trained student the one flat buffer; for a LoRA student
(``lora.py``, train.py:180-202) the two saved modules in bf16 plus the adapter's fp32 masters (``sd_adamw_f32_shadow``: fp32
moments, and the bf16 operands of the merge / projection kernels written in the same pass).  ``model.finalize_grads()`` runs
first -- that is where a LoRA student projects the accumulated weight gradient onto its adapter.
"""
import torch

from . import _lib, ops


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip=0.0):
        self.model = model
        params = [p for p in model.parameters() if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, clip=clip))
        # Moments: one flat buffer per kind, laid out in the order of model.optim_segments() -- for the fully trained
        # student that is the model's own flat layout; a LoRA student (lora.py) has its two saved modules in bf16 and
        # the adapter's fp32 masters.
        dev = model.flat.device
        segs = model.optim_segments()
        n16 = sum(p.numel() for kind, p, *_ in segs if kind == "bf16")
        n32 = sum(p.numel() for kind, p, *_ in segs if kind == "f32_shadow")
        self.exp_avg = torch.zeros(n16, dtype=torch.bfloat16, device=dev)
        self.exp_avg_sq = torch.zeros(n16, dtype=torch.bfloat16, device=dev)
        self.exp_avg32 = torch.zeros(n32, dtype=torch.float32, device=dev) if n32 else None
        self.exp_avg_sq32 = torch.zeros(n32, dtype=torch.float32, device=dev) if n32 else None
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._partials = torch.empty(_lib.SUMSQ_PARTIALS, dtype=torch.float32, device=dev)  # sd_sumsq_bf16 scratch, this optimizer's own
        self._step = 0
        self._measured_clip = None  # set by grad_norm() for the following step()

    def _reduce_sumsq(self):
        self.model.finalize_grads()     # (a LoRA student projects dW onto its adapter here)
        self._sumsq.zero_()
        for _, _, g, _, _ in self.model.optim_segments():
            ops.sumsq(g, self._sumsq, self._partials)

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        m = self.model
        if m.flat_grad is None:
            return None
        self._step += 1
        clip = float(g["clip"])
        ss = None
        if self._measured_clip is not None:   # grad_norm() already reduced THIS gradient: the kernel clips against it
            clip, ss, self._measured_clip = self._measured_clip, self._sumsq, None
        elif clip > 0:
            self._reduce_sumsq()
            ss = self._sumsq
        else:
            m.finalize_grads()
        b1, b2 = g["betas"]
        wd = float(g["weight_decay"])
        o16 = o32 = 0
        for kind, p, gr, is_mat, extra in m.optim_segments(split_decay=wd != 0.0):
            n = p.numel()
            if kind == "bf16":
                ops.adamw_(p, gr, self.exp_avg[o16:o16 + n], self.exp_avg_sq[o16:o16 + n], float(g["lr"]), b1, b2, g["eps"],
                           wd if is_mat else 0.0, self._step, ss, clip)
                o16 += n
            else:
                shadow, shadow_scaled, scale, owner = extra
                ops.adamw_f32_shadow_(p, gr, self.exp_avg32[o32:o32 + n], self.exp_avg_sq32[o32:o32 + n], shadow,
                                      shadow_scaled, scale, float(g["lr"]), b1, b2, g["eps"], wd if is_mat else 0.0,
                                      self._step, ss, clip)
                owner.mark_updated()
                o32 += n
        return None

    @torch.no_grad()
    def grad_norm(self, max_norm):
        """L2 norm of the whole gradient (a 0-d fp32 tensor, no host sync) = what ``clip_grad_norm_`` returns (HF
        trainer.py:2535-2539); the next ``step()`` applies min(1, max_norm / (norm + 1e-6)) to the gradient INSIDE the
        fused update instead of rewriting the 1.2 GB gradient buffer first (``.grad`` itself stays unclipped).
        max_norm = inf (HF asks that way for the norm alone) or <= 0: measure only."""
        self._reduce_sumsq()
        self._measured_clip = float(max_norm) if 0 < max_norm < float("inf") else 0.0
        return self._sumsq.sqrt().squeeze(0)

    def state_dict(self):
        """HF Trainer checkpoints ``optimizer.state_dict()`` (optimizer.pt): the flat moments + step count."""
        sd = super().state_dict()
        sd["flat"] = {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": self._step}
        if self.exp_avg32 is not None:
            sd["flat"].update(exp_avg32=self.exp_avg32, exp_avg_sq32=self.exp_avg_sq32)
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        flat = state_dict.pop("flat", None)
        super().load_state_dict(state_dict)
        if flat is not None:
            self.exp_avg.copy_(flat["exp_avg"])
            self.exp_avg_sq.copy_(flat["exp_avg_sq"])
            if self.exp_avg32 is not None:
                self.exp_avg32.copy_(flat["exp_avg32"])
                self.exp_avg_sq32.copy_(flat["exp_avg_sq32"])
            self._step = int(flat["step"])

    def last_grad_norm(self):
        """Global gradient norm seen by the last step / measurement (device tensor)."""
        return self._sumsq.sqrt()
