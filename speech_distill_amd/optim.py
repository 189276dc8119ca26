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
"""
import torch

from . import _lib, ops


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip=0.0):
        self.model = model
        params = [p for p in model.parameters() if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, clip=clip))
        self.exp_avg = torch.zeros_like(model.flat)
        self.exp_avg_sq = torch.zeros_like(model.flat)
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=model.flat.device)
        self._partials = torch.empty(_lib.SUMSQ_PARTIALS, dtype=torch.float32, device=model.flat.device)  # sd_sumsq_bf16 scratch, this optimizer's own
        self._step = 0
        self._measured_clip = None  # set by grad_norm() for the following step()
        # contiguous runs of matrices / gains in the flat layout (for decay on matrices only)
        runs, cur = [], None
        for name, (o, n, shape) in model._slices.items():
            is_mat = len(shape) == 2
            n8 = (n + 7) // 8 * 8
            if cur is not None and cur[2] == is_mat and cur[1] == o:
                cur[1] = o + n8
            else:
                cur = [o, o + n8, is_mat]
                runs.append(cur)
        self._runs = [(a, b, m) for a, b, m in runs]

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
            self._sumsq.zero_()
            ops.sumsq(m.flat_grad, self._sumsq, self._partials)
            ss = self._sumsq
        b1, b2 = g["betas"]
        wd = float(g["weight_decay"])
        spans = [(0, m.numel_flat, True)] if wd == 0.0 else self._runs
        for a, b, is_mat in spans:
            ops.adamw_(m.flat[a:b], m.flat_grad[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b], float(g["lr"]), b1, b2,
                       g["eps"], wd if is_mat else 0.0, self._step, ss, clip)
        return None

    @torch.no_grad()
    def grad_norm(self, max_norm):
        """L2 norm of the whole gradient (a 0-d fp32 tensor, no host sync) = what ``clip_grad_norm_`` returns (HF
        trainer.py:2535-2539); the next ``step()`` applies min(1, max_norm / (norm + 1e-6)) to the gradient INSIDE the
        fused update instead of rewriting the 1.2 GB gradient buffer first (``.grad`` itself stays unclipped).
        max_norm = inf (HF asks that way for the norm alone) or <= 0: measure only."""
        self._sumsq.zero_()
        ops.sumsq(self.model.flat_grad, self._sumsq, self._partials)
        self._measured_clip = float(max_norm) if 0 < max_norm < float("inf") else 0.0
        return self._sumsq.sqrt().squeeze(0)

    def state_dict(self):
        """HF Trainer checkpoints ``optimizer.state_dict()`` (optimizer.pt): the flat bf16 moments + step count."""
        sd = super().state_dict()
        sd["flat"] = {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": self._step}
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        flat = state_dict.pop("flat", None)
        super().load_state_dict(state_dict)
        if flat is not None:
            self.exp_avg.copy_(flat["exp_avg"])
            self.exp_avg_sq.copy_(flat["exp_avg_sq"])
            self._step = int(flat["step"])

    def last_grad_norm(self):
        """Global gradient norm seen by the last step / measurement (device tensor)."""
        return self._sumsq.sqrt()
