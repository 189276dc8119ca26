"""Data-parallel gradient averaging for the flat bf16 gradient buffer (one process per GPU).

What the reference gets implicitly from accelerate -> torch DDP when launched under torchrun
(HF trainer.py:1615-1626, no_sync at :1757): a bucketed all-reduce(avg) of the student's gradients
overlapped with backward.  Here the buckets are SLICES of the model's flat gradient buffer, closed
in the order backward finishes them -- head/final-norm, layer L-1 .. 0, embedding -- and each
bucket's all-reduce is issued on a dedicated communication stream the moment the C runner reports
that the kernels producing it are enqueued (``sd_stage_cb``), so it runs under the remaining
backward kernels.  ``torch.distributed`` backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used
by the CPU tests of the bucketing logic.  The frozen teacher is replicated and never communicated.

xGMI is point-to-point (7 links per GPU): per-layer buckets (31.5 MB for the 0.6B student) are
large enough for RCCL to spread over all links; fusing more layers per bucket is a knob
(``layers_per_bucket``).
"""
from __future__ import annotations

import contextlib

import torch
import torch.distributed as dist

STAGE_HEAD, STAGE_EMBED = -1, -2


def bucket_plan(layer_ranges, embed_range, norm_range, numel, layers_per_bucket=1):
    """-> list of (stage_that_closes_it, start, end) in backward completion order.

    The tied embedding / lm_head gradient is only final after the embedding scatter-add
    (STAGE_EMBED), the final-norm gain after STAGE_HEAD."""
    plan = [(STAGE_HEAD, norm_range[0], norm_range[1])]
    L = len(layer_ranges)
    l = L - 1
    while l >= 0:
        lo = max(0, l - layers_per_bucket + 1)
        plan.append((lo, layer_ranges[lo][0], layer_ranges[l][1]))
        l = lo - 1
    plan.append((STAGE_EMBED, embed_range[0], embed_range[1]))
    if numel > norm_range[1]:  # untied lm_head lives after the final norm; final once the head stage is done
        plan.insert(1, (STAGE_HEAD, norm_range[1], numel))
    return plan


class FlatGradAllReduce:
    """Averages ``flat_grad`` over the process group, bucket by bucket, overlapped with backward."""

    def __init__(self, flat_grad_getter, plan, group=None, comm_stream=None):
        self._get = flat_grad_getter
        self.plan = plan
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.sync = True
        self.cuda = torch.cuda.is_available()
        self.comm = comm_stream if comm_stream is not None else (torch.cuda.Stream() if self.cuda else None)
        self._works = []
        self.issued = []  # (stage, start, end) actually reduced in the current step (for tests / stats)

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation: skip communication on non-final micro-batches (HF trainer.py:1757)."""
        old, self.sync = self.sync, False
        try:
            yield
        finally:
            self.sync = old

    def begin_step(self):
        self._works, self.issued = [], []

    def on_stage(self, stage):
        """Host callback from the backward runner: grads of `stage` are enqueued on the compute stream."""
        if not self.sync or self.world == 1:
            return
        flat = self._get()
        for (st, a, b) in self.plan:
            if st != stage:
                continue
            chunk = flat[a:b]
            if self.cuda:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                self.comm.wait_event(ev)
                with torch.cuda.stream(self.comm):
                    dist.all_reduce(chunk, op=dist.ReduceOp.AVG, group=self.group)
            else:  # gloo has no AVG: sum then scale
                w = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self._works.append((w, chunk))
            self.issued.append((st, a, b))

    def finish(self):
        """Order everything after the last all-reduce (stream wait on GPU, no host sync)."""
        if self.cuda and self.comm is not None:
            torch.cuda.current_stream().wait_stream(self.comm)
        for w, chunk in self._works:
            w.wait()
            chunk.div_(self.world)
        self._works = []


def attach(model, group=None, layers_per_bucket=1):
    """Wire a HipQwen3ForCausalLM's backward stage callback to overlapped all-reduces."""
    plan = bucket_plan(model.layer_ranges, model.embed_range, model.norm_range, model.numel_flat, layers_per_bucket)
    red = FlatGradAllReduce(lambda: model.flat_grad, plan, group)
    model._stage_cb = red.on_stage
    model._reducer = red
    model.no_sync = red.no_sync
    return red
