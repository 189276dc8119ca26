"""Data-parallel gradient averaging for the flat bf16 gradient buffer (one process per GPU).

What the reference gets implicitly from accelerate -> torch DDP when launched under torchrun
(HF trainer.py:1615-1626, no_sync at :1757): a bucketed all-reduce(avg) of the student's gradients
overlapped with backward.  Here the buckets are SLICES of the model's flat gradient buffer, closed
in the order backward finishes them -- head/final-norm, layer L-1 .. 0, embedding -- and each
bucket's all-reduce is issued on a dedicated communication stream the moment the C runner reports
that the kernels producing it are enqueued (``sd_stage_cb``), so it runs under the remaining
backward kernels.  ``torch.distributed`` backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used
by the CPU tests of the bucketing logic.  The frozen teacher is replicated and never communicated.

xGMI is point-to-point (7 links x ~153 GB/s per GPU, no switch): a RING all-reduce of a bucket of S bytes moves
2*(W-1)/W*S per GPU over ONE link per direction, a DIRECT reduce-scatter + all-gather moves (W-1)/W*S/(W-1) = S/W per
link over all W-1 links at once (13.8 ms vs ~2 ms for the 1.21 GB of student gradients at W = 8).
Which of the two RCCL picks for ``all_reduce`` is its own choice, so both collective forms are offered -- ``algo=
"allreduce"`` (one ``all_reduce(AVG)`` per bucket) and ``algo="rs_ag"`` (in-place ``reduce_scatter_tensor(AVG)`` +
``all_gather_into_tensor`` on the same slice) -- together with the bucket size (``layers_per_bucket``: 31.5 MB per layer
for the 0.6B student) and a per-bucket timing mode (``timing=True``: events on the communication stream + the exposed
wait in ``finish``), so that a run on real links can tell them apart (``bench.py --comm-algo / --layers-per-bucket``).

Status of the two forms: ``allreduce`` is the default and the only one that has run with more than one rank anywhere
(gloo ranks on CPU / one GPU; the RCCL path itself only as a one-rank rehearsal).  ``rs_ag`` is EXPERIMENTAL until
``bench.py``'s ``grad_sync_ok`` and a parameter checksum have passed on a real multi-GPU node: its in-place
reduce-scatter into a slice of its own input, the AVG semantics of ``reduce_scatter_tensor`` and the bit-identity of the
gathered gradients across ranks are unverified at W > 1.  A bucket whose size does not divide by W cannot use it and
falls back to ``all_reduce`` -- loudly (one warning per bucket size, counted in ``stats["rs_ag_fallbacks"]``).
"""
from __future__ import annotations

import contextlib
import warnings

import torch
import torch.distributed as dist

STAGE_HEAD, STAGE_EMBED = -1, -2


def bucket_plan(layer_ranges, embed_range, norm_range, numel, layers_per_bucket=1, split_embedding=False):
    """-> list of (stage_that_closes_it, start, end) in backward completion order.

    The tied embedding / lm_head gradient (27 % of the student: 327 MB) is only final after the embedding
    scatter-add (STAGE_EMBED), i.e. at the very end of backward.  With ``split_embedding`` its dense
    lm_head part is instead reduced right after STAGE_HEAD -- under the whole layer backward -- and the
    scatter part is exchanged as rows (FlatGradAllReduce._exchange_embedding_rows)."""
    plan = [(STAGE_HEAD, norm_range[0], norm_range[1])]
    if split_embedding:
        plan.append((STAGE_HEAD, embed_range[0], embed_range[1]))
    L = len(layer_ranges)
    l = L - 1
    while l >= 0:
        lo = max(0, l - layers_per_bucket + 1)
        plan.append((lo, layer_ranges[lo][0], layer_ranges[l][1]))
        l = lo - 1
    if not split_embedding:
        plan.append((STAGE_EMBED, embed_range[0], embed_range[1]))
    if numel > norm_range[1]:  # untied lm_head lives after the final norm; final once the head stage is done
        plan.insert(1, (STAGE_HEAD, norm_range[1], numel))
    return plan


class FlatGradAllReduce:
    """Averages ``flat_grad`` over the process group, bucket by bucket, overlapped with backward."""

    def __init__(self, flat_grad_getter, plan, group=None, comm_stream=None, split_embedding=False, algo="allreduce",
                 timing=False, rehearse_single_rank=False):
        if algo not in ("allreduce", "rs_ag"):
            raise ValueError(f"algo {algo!r}: 'allreduce' or 'rs_ag'")
        self._get = flat_grad_getter
        self.plan = plan
        self.split_embedding = split_embedding
        self._emb = None
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # a one-rank process group issues every collective all the same (RCCL runs them as local copies): the
        # rehearsal of the N > 1 call sequence on a one-GPU box (bench.py --force-dist, tests/test_00_gpu_rccl1.py)
        self.active = self.world > 1 or (rehearse_single_rank and dist.is_initialized())
        self.algo = algo
        self._warned_sizes = set()
        self.timing = timing
        self._timed = []      # (stage, numel, start event, end event) of the current step
        self._exposed = None  # (event before, event after) the main stream's wait for the communication stream
        self.sync = True
        backend = dist.get_backend(group) if dist.is_initialized() else "none"
        # RCCL ("nccl"): in-stream all_reduce(AVG) on a dedicated communication stream.  gloo (CPU tests,
        # single-GPU rehearsal): no AVG and no stream semantics -> async SUM, scaled in finish().
        self.cuda = torch.cuda.is_available() and backend == "nccl"
        self.comm = comm_stream
        if self.comm is None and self.cuda:
            from . import ops  # a stream measured to overlap the main and the weight-gradient streams (ops.concurrent_stream)
            self.comm = ops.concurrent_stream(torch.device("cuda", torch.cuda.current_device()), "comm")
        self._works = []
        self.issued = []  # (stage, start, end) actually reduced in the current step (for tests / stats)
        self.stats = {"backwards": 0, "synced": 0}  # backward passes seen / of which communicated

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
        self._emb = None
        self._timed, self._exposed = [], None
        self.stats["backwards"] += 1
        self.stats["synced"] += int(self.sync and self.active)

    def wants_split_embedding(self):
        return self.split_embedding and self.sync and self.active

    def set_embedding_exchange(self, ids, dx0, embed_grad):
        """Called by the model before backward: token ids [M], buffer that will receive d loss / d embedding
        output [M,h], and the [V,h] gradient view the rows are scattered into.

        Ranks may hold different M = B*T (the collator pads to the per-batch maximum, data.py:280-327), so the row
        counts are exchanged here -- on the idle communication stream, before the backward kernels are enqueued, so the
        host read does not wait for compute -- and ``_exchange_embedding_rows`` pads every rank to the largest."""
        self._emb = (ids, dx0, embed_grad)
        m_local = ids.numel()
        ctx = torch.cuda.stream(self.comm) if self.cuda else contextlib.nullcontext()
        with ctx:
            mine = torch.tensor([m_local], dtype=torch.int64, device=ids.device)
            counts = [torch.zeros_like(mine) for _ in range(self.world)]
            dist.all_gather(counts, mine, group=self.group)
            self._counts = [int(c) for c in torch.cat(counts).tolist()]

    def _exchange_embedding_rows(self):
        """All-gather (ids, rows) of every rank, then the same deterministic scatter-add on every rank:
        dE += (1/W) sum_r scatter(ids_r, rows_r).  Runs after the dense part's all-reduce (same stream).
        Rows beyond a rank's own count are padding (zeros) and are not scattered."""
        ids, dx0, embed_grad = self._emb
        W, counts = self.world, self._counts
        m_local, m_max = ids.numel(), max(counts)
        if m_local < m_max:
            ids = torch.cat([ids, ids.new_zeros(m_max - m_local)])
            dx0 = torch.cat([dx0, dx0.new_zeros(m_max - m_local, dx0.shape[1])])
        all_ids = [torch.empty_like(ids) for _ in range(W)]
        all_rows = [torch.empty_like(dx0) for _ in range(W)]
        dist.all_gather(all_ids, ids.contiguous(), group=self.group)
        dist.all_gather(all_rows, dx0.contiguous(), group=self.group)
        for r in range(W):
            self._scatter_rows(all_ids[r][:counts[r]], all_rows[r][:counts[r]], embed_grad, 1.0 / W)

    @staticmethod
    def _scatter_rows(ids, rows, embed_grad, scale):
        from . import ops
        ops.embedding_bwd(ids.contiguous(), rows.contiguous(), embed_grad, scale=scale)

    def on_stage(self, stage):
        """Host callback from the backward runner: grads of `stage` are enqueued on the compute stream."""
        if not self.sync or not self.active:
            return
        if stage == STAGE_EMBED and self._emb is not None:
            if self.cuda:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                self.comm.wait_event(ev)
                with torch.cuda.stream(self.comm):
                    t0 = self._tick()
                    self._exchange_embedding_rows()
                    self._tock(stage, self._emb[1].numel(), t0)
            else:
                if self._emb[1].is_cuda:
                    torch.cuda.current_stream().synchronize()
                for w, chunk in self._works:  # the dense part must be averaged before rows are added
                    w.wait()
                    chunk.div_(self.world)
                self._works = []
                self._exchange_embedding_rows()
            self.issued.append((stage, -1, -1))
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
                    t0 = self._tick()
                    self._reduce_on_stream(chunk)
                    self._tock(st, chunk.numel(), t0)
            else:  # gloo has no AVG: sum then scale (device tensors: wait for the producing kernels first)
                if chunk.is_cuda:
                    torch.cuda.current_stream().synchronize()
                w = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self._works.append((w, chunk))
            self.issued.append((st, a, b))

    def _reduce_on_stream(self, chunk):
        """Average one bucket over the group on the current (communication) stream, in place."""
        W = self.world
        if self.algo == "rs_ag" and chunk.numel() % W == 0:
            # every tensor of the flat layout starts on a multiple of 8 elements, so buckets divide by W = 2, 4, 8.
            # In place: rank r's shard of the sum lands in its own slot, then every rank gathers all slots.
            n = chunk.numel() // W
            r = dist.get_rank(self.group)
            mine = chunk[r * n:(r + 1) * n]
            dist.reduce_scatter_tensor(mine, chunk, op=dist.ReduceOp.AVG, group=self.group)
            dist.all_gather_into_tensor(chunk, mine, group=self.group)
        else:
            if self.algo == "rs_ag":  # asked for the direct form, cannot have it: say so (once per bucket size)
                self.stats["rs_ag_fallbacks"] = self.stats.get("rs_ag_fallbacks", 0) + 1
                if chunk.numel() not in self._warned_sizes:
                    self._warned_sizes.add(chunk.numel())
                    warnings.warn(f"FlatGradAllReduce: algo='rs_ag' but a bucket of {chunk.numel()} elements does not divide "
                                  f"by the world size {W}: this bucket uses all_reduce (ring or direct is then RCCL's choice)")
            dist.all_reduce(chunk, op=dist.ReduceOp.AVG, group=self.group)

    def _tick(self):
        if not self.timing:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def _tock(self, stage, numel, t0):
        if t0 is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._timed.append((stage, numel, t0, e))

    def timings(self):
        """After a device synchronisation: ([(stage, bytes, ms on the communication stream)], exposed ms) of the last
        step -- exposed = how long the compute stream sat in ``finish`` waiting for the communication stream."""
        per = [(st, 2 * n, t0.elapsed_time(t1)) for st, n, t0, t1 in self._timed]
        exposed = self._exposed[0].elapsed_time(self._exposed[1]) if self._exposed else None
        return per, exposed

    def finish(self):
        """Order everything after the last all-reduce (stream wait on GPU, no host sync)."""
        if self.cuda and self.comm is not None:
            if self.timing and self.sync and self.active:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(torch.cuda.current_stream())
                torch.cuda.current_stream().wait_stream(self.comm)
                b.record(torch.cuda.current_stream())
                self._exposed = (a, b)
            else:
                torch.cuda.current_stream().wait_stream(self.comm)
        for w, chunk in self._works:
            w.wait()
            chunk.div_(self.world)
        self._works = []


def attach(model, group=None, layers_per_bucket=None, split_embedding=True, algo=None, timing=False,
           rehearse_single_rank=False):
    """Wire the backward stage callback of a flat-gradient model (HipQwen3ForCausalLM) to overlapped all-reduces.
    ``layers_per_bucket`` / ``algo`` default to the environment variables SD_LAYERS_PER_BUCKET (1) / SD_COMM_ALGO
    ("allreduce"; "rs_ag" is EXPERIMENTAL, see the module docstring), so a ``torchrun scripts/train.py`` launch can A/B
    them without code changes."""
    import os
    layers_per_bucket = int(os.environ.get("SD_LAYERS_PER_BUCKET", "1")) if layers_per_bucket is None else layers_per_bucket
    algo = os.environ.get("SD_COMM_ALGO", "allreduce") if algo is None else algo
    split_embedding = split_embedding and model.dims.tie_word_embeddings
    plan = bucket_plan(model.layer_ranges, model.embed_range, model.norm_range, model.numel_flat, layers_per_bucket,
                       split_embedding)
    red = FlatGradAllReduce(lambda: model.flat_grad, plan, group, split_embedding=split_embedding, algo=algo, timing=timing,
                            rehearse_single_rank=rehearse_single_rank)
    model._stage_cb = red.on_stage
    model._reducer = red
    model.no_sync = red.no_sync
    return red


def speaks_flat_grad(model):
    """True for models that own their gradient exchange through this module instead of torch DDP's autograd hooks:
    one flat gradient buffer + a backward runner that reports finished stages (HipQwen3ForCausalLM)."""
    return all(hasattr(model, a) for a in ("flat_grad", "layer_ranges", "embed_range", "norm_range", "numel_flat",
                                           "_stage_cb"))


class HipDataParallel(torch.nn.Module):
    """What ``torch.nn.parallel.DistributedDataParallel`` is to an autograd model, for a flat-gradient model.

    The reference gets data parallelism from accelerate, which wraps the student in torch DDP inside HF Trainer
    (HF trainer.py:1615-1626) and skips the all-reduce on accumulation micro-batches through ``model.no_sync()``
    (HF trainer.py:1757).  torch DDP cannot serve the HIP student: its reducer is driven by autograd hooks on the
    parameters, and the backward runner writes the flat gradient buffer directly.  This wrapper is what
    ``DistillationTrainer._wrap_model`` hands to the training loop instead: same ``.module`` / ``no_sync()`` surface
    (accelerate's ``no_sync(model)`` finds it by duck typing), parameters broadcast from rank 0 at construction as
    DDP does, gradient averaging by ``FlatGradAllReduce`` under the backward.  With one process it only forwards.
    """

    def __init__(self, module, group=None, layers_per_bucket=None, split_embedding=True, broadcast_params=True):
        super().__init__()
        self.module = module
        self.reducer = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            if broadcast_params and getattr(module, "flat", None) is not None:
                src = dist.get_global_rank(group, 0) if group is not None else 0
                dist.broadcast(module.flat, src=src, group=group)
                lora = getattr(module, "_lora", None)
                if lora is not None:   # a LoRA student: rank 0's adapter and residual base too (lora.py)
                    dist.broadcast(lora.master, src=src, group=group)
                    dist.broadcast(lora.base, src=src, group=group)
                    lora.refresh_shadows()
            self.reducer = attach(module, group, layers_per_bucket, split_embedding)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def no_sync(self):
        return self.reducer.no_sync() if self.reducer is not None else contextlib.nullcontext()

    def zero_grad(self, set_to_none: bool = True):
        return self.module.zero_grad(set_to_none)

    # what Trainer reads from the object it trains
    @property
    def config(self):
        return self.module.config

    def gradient_checkpointing_enable(self, *a, **k):
        return self.module.gradient_checkpointing_enable(*a, **k)


def unwrap(model):
    """The module under HipDataParallel / torch DDP / DataParallel (the model itself otherwise)."""
    while isinstance(model, (HipDataParallel, torch.nn.parallel.DistributedDataParallel, torch.nn.DataParallel)):
        model = model.module
    return model
