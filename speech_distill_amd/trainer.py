"""Drop-in for the reference's ``DistillationTrainer`` (train.py:24-116): same constructor keywords,
same ``compute_loss`` contract, same three logged sub-losses, the HIP kernels underneath.

Step body (train.py:43-116):  pop the teacher / extra keys  ->  student forward (labels still in the
inputs, as in the reference)  ->  pop labels  ->  teacher no-grad forward unless pre-extracted top-K is
present  ->  on-the-fly log-softmax + top-K unless the teacher is quantized or top_k <= 0  ->
DistillationLoss  ->  log {student_loss, teacher_loss, distill_loss} when
``state.global_step % args.logging_steps == 0``  ->  ``loss`` or ``(loss, outputs)``.
"""
import os
import time

import torch
from transformers import Trainer

from . import ddp, ops
from .distillation_loss import DistillationLoss
from .qwen3 import HipQwen3ForCausalLM


class DistillationTrainer(Trainer):
    def __init__(self, *args, teacher_model=None, temperature=2.0, alpha=0.5, top_k=100, is_quantized_teacher=False,
                 **kwargs):
        super().__init__(*args, **kwargs)
        self.teacher_model = teacher_model
        if self.teacher_model is not None:
            self.teacher_model.eval()
        self.top_k = top_k
        self.distill_loss_fn = DistillationLoss(temperature=temperature, alpha=alpha)
        self.is_quantized_teacher = is_quantized_teacher
        # compute back-ends; the product ones are the HIP kernels (tests may inject a checker here)
        self._extract_topk = ops.logsoftmax_topk
        # throughput of the loop: positions (B*T of every training micro-batch, padding included -- what the kernels
        # process and what bench.py counts) seen by THIS rank; `log` turns it into whole-job tokens/sec
        self.tokens_seen = 0
        self._tps_mark = None  # (perf_counter, tokens_seen) at the previous training log
        # The frozen teacher does not depend on the student: on GPUs its forward (+ top-K) runs on a second HIP
        # stream beside the student forward (+4.5 % step throughput on MI355X); results are identical.
        self.overlap_teacher = True
        self._teacher_stream = None
        self._rows_ahead = {}  # id(labels) -> (key, (rows, row_labels), event) selected in get_batch_samples for this optimizer step
        self._copy_stream = None
        # The teacher's pass of micro-batch i+1 of an accumulation window is enqueued at the END of training_step(i): HF's
        # per-micro-batch host read (logging_nan_inf_filter, on by default) drains the main stream after every backward,
        # and the GPU then has the next teacher pass to run while the host walks back to compute_loss (+1.6-2.1 % loop
        # throughput; with HF's read off the host is ahead anyway and it measured -0.5 %, so "auto" follows that flag).
        # SD_TEACHER_AHEAD=0 / 1 forces it.
        self._window, self._window_next, self._ahead_results, self._prefetched = [], [], {}, None
        # Training steps apply both lm_heads, the top-K and the loss only to the rows the loss reads (positions whose
        # NEXT label is not -100, distillation_loss.py:31-45) instead of computing all B*T rows and masking them
        # afterwards; loss and gradients are the same, the head is ~(masked fraction) cheaper.  Needs one host sync
        # per step for the row count.  Off: the full [B,T,V] path; return_outputs=True always uses the full path.
        self.compact_head = True
        self._hip_dp = None
        # The optimizer HF would build by default (AdamW over ~310 HF-named tensors, HF trainer.py:1778-1796) is replaced
        # by the one-launch FlatAdamW over the flat buffers when the student offers them; set False to keep HF's.
        self.fused_optimizer = True

    # ------------------------------------------------------------------------------ HF Trainer plumbing
    def _wrap_model(self, model, training=True, dataloader=None):
        """HF trainer.py:1602-1626: a model this hook leaves unwrapped goes through ``accelerator.prepare``, which under
        torchrun wraps it in torch DDP (the reference's data parallelism) and under ``bf16=True`` wraps ``forward`` to
        copy every bf16 output to fp32.  Neither fits a flat-gradient model (see ``ddp.HipDataParallel``): its
        gradients never pass autograd hooks, and its [.,V] logits are consumed as bf16 by the loss kernel.  Such a
        model is handed to the loop inside ``HipDataParallel`` instead -- the one owner of gradient averaging, with the
        ``no_sync()`` accelerate looks for on accumulation micro-batches; HF then prepares only the optimizer."""
        if isinstance(model, ddp.HipDataParallel):
            return model
        if ddp.speaks_flat_grad(model):
            if self._hip_dp is None or self._hip_dp.module is not model:
                self._hip_dp = ddp.HipDataParallel(model)
            return self._hip_dp
        return super()._wrap_model(model, training=training, dataloader=dataloader)

    def log(self, logs, *args, **kwargs):
        """train.py:107-114 logs the three sub-losses only; SURVEY section 8 f-1 asks the counterpart for tokens/sec.  HF's
        own training log (the dict with ``loss`` / ``learning_rate``, every ``logging_steps`` optimizer steps, after a
        ``.item()`` that has drained the GPU) gets ``tokens_per_second``: positions of all ranks' micro-batches since the
        previous such log over the wall time between the two."""
        if "loss" in logs and "learning_rate" in logs and self._tps_mark is not None:
            now = time.perf_counter()
            t0, n0 = self._tps_mark
            if now > t0 and self.tokens_seen > n0:
                logs = dict(logs)
                logs["tokens_per_second"] = round((self.tokens_seen - n0) * max(1, self.args.world_size) / (now - t0), 1)
            self._tps_mark = (now, self.tokens_seen)
        return super().log(logs, *args, **kwargs)

    def create_optimizer(self, model=None):
        """HF trainer.py `create_optimizer(self, model=None)`: TrainingArguments' default is torch AdamW (the reference passes no
        ``optim``, train.py:331-354).  For a flat-buffer student on the GPU the same update -- same hyper-parameters,
        weight decay on matrices only like HF's parameter groups, moments in the parameters' dtype (quirk Q5) -- is ONE
        fused launch over the flat parameter / gradient / moment buffers (``sd_adamw_bf16``).  HF keeps clipping
        (``max_grad_norm``) and the learning-rate schedule: the scheduler writes ``param_groups[0]["lr"]``."""
        core = ddp.unwrap(self.model)
        default_adamw = str(getattr(self.args.optim, "value", self.args.optim)).startswith("adamw_torch")
        if (self.optimizer is None and self.fused_optimizer and default_adamw and isinstance(core, HipQwen3ForCausalLM)
                and core.flat.is_cuda and (core._lora is not None or all(p.requires_grad for p in core.parameters()))):
            from .optim import FlatAdamW
            self.optimizer = FlatAdamW(core, lr=self.args.learning_rate, betas=(self.args.adam_beta1, self.args.adam_beta2),
                                       eps=self.args.adam_epsilon, weight_decay=self.args.weight_decay)
            return self.optimizer
        if isinstance(core, HipQwen3ForCausalLM) and core._lora is not None and self.optimizer is None:
            raise NotImplementedError("a LoRA student (lora.py) trains through FlatAdamW only: its adapter gradients are "
                                      "bf16 projections of the flat weight gradient next to fp32 masters; leave "
                                      "TrainingArguments.optim at its adamw_torch default, as the reference does")
        # (HF calls create_optimizer(model) on its delay_optimizer_creation path: FSDP / SageMaker model parallel)
        return super().create_optimizer(model) if model is not None else super().create_optimizer()

    def _clip_grad_norm(self, model):
        """HF trainer.py:2535-2539 calls ``clip_grad_norm_`` over ~310 parameter tensors (10 ms of host time per
        optimizer step with the GPU idle).  With the flat buffers the norm is one reduction, and FlatAdamW folds the
        clip coefficient into its update; the returned pre-clip norm is the same number HF logs as ``grad_norm``."""
        core = ddp.unwrap(model)
        opt = getattr(self.optimizer, "optimizer", self.optimizer)
        if isinstance(core, HipQwen3ForCausalLM) and core.flat_grad is not None and core.flat_grad.is_cuda:
            from .optim import FlatAdamW
            if isinstance(opt, FlatAdamW) and opt.model is core:
                return opt.grad_norm(self.args.max_grad_norm)
            core.finalize_grads()
            ss = torch.zeros(1, dtype=torch.float32, device=core.flat_grad.device)
            ops.sumsq(core.flat_grad, ss)
            norm = ss.sqrt().squeeze(0)
            core.flat_grad.mul_((self.args.max_grad_norm / (norm + 1e-6)).clamp(max=1.0).to(core.flat_grad.dtype))
            return norm
        return super()._clip_grad_norm(model)

    def _save(self, output_dir=None, state_dict=None):
        """HF trainer.py `_save`: a model that is not a PreTrainedModel gets a bare ``model.safetensors``.  The
        reference's checkpoints (save_strategy="epoch", train.py:341-345) are ``save_pretrained`` directories
        (``config.json`` + weights, tied head left out) -- write the same for a model that offers it."""
        core = ddp.unwrap(self.model)
        if not hasattr(core, "save_pretrained") or isinstance(core, _pretrained_types()):
            return super()._save(output_dir, state_dict)
        output_dir = output_dir if output_dir is not None else self.args.output_dir
        os.makedirs(output_dir, exist_ok=True)
        core.save_pretrained(output_dir, state_dict=state_dict)
        if self.processing_class is not None and hasattr(self.processing_class, "save_pretrained"):
            self.processing_class.save_pretrained(output_dir)
        torch.save(self.args, os.path.join(output_dir, "training_args.bin"))

    @staticmethod
    def _rows_key(lab, speech_mask, am, tam):
        return (id(lab), lab.data_ptr(), tuple(lab.shape), None if speech_mask is None else speech_mask.data_ptr(),
                None if am is None else am.data_ptr(), None if tam is None else tam.data_ptr())

    def get_batch_samples(self, epoch_iterator, num_batches, device):
        """HF fetches the micro-batches of one optimizer step here, together, right after it has enqueued the previous
        optimizer step.  Two things are done with that (``SD_ROWS_AHEAD=0`` turns both off):

        * the loss-row selection of every micro-batch (one host read each: the row count) happens now, so the
          accumulation window itself has no host reads of ours -- with ``logging_nan_inf_filter=False`` (HF's own
          per-micro-batch read) the host enqueues the whole window ahead of the GPU;
        * the fetch -- accelerate's host-to-device copies, HF's token count, the row selection -- runs on a stream of its
          own ("h2d"): none of it depends on the optimizer kernels still queued on the main stream, and the frozen
          teacher's pass in ``compute_loss`` then waits for THAT stream's event only, i.e. the teacher's forward of the
          window's first micro-batch runs beside the AdamW update (HBM-bound) instead of after it.

        Same selection, validation and arguments as ``compute_loss`` would use on its own."""
        core = ddp.unwrap(self.model)
        ahead = (self.compact_head and os.environ.get("SD_ROWS_AHEAD", "1") != "0" and isinstance(core, HipQwen3ForCausalLM)
                 and core.flat.is_cuda and isinstance(self.distill_loss_fn, DistillationLoss))
        if not ahead:
            self._rows_ahead.clear()
            self._window, self._window_next, self._ahead_results, self._prefetched = [], [], {}, None
            return super().get_batch_samples(epoch_iterator, num_batches, device)
        # window k comes from the previous call's lookahead when there is one (same iterator object = same epoch)
        pre, self._prefetched = self._prefetched, None
        if pre is not None and pre[0] is epoch_iterator:
            batch_samples, num_items, selected, ready = pre[1]
        else:
            batch_samples, num_items, selected, ready = self._fetch_window(core, epoch_iterator, num_batches, device)
        main = torch.cuda.current_stream(core.flat.device)
        main.wait_event(ready)
        live = {id(b.get("labels")) for b in batch_samples if isinstance(b, dict)}
        self._rows_ahead = {k: v for k, v in self._rows_ahead.items() if k in live}
        self._ahead_results = {k: v for k, v in self._ahead_results.items() if k in live}
        for key, rr in selected:
            self._rows_ahead[key[0]] = (key, rr, ready)
        self._window, self._window_next = list(batch_samples), []
        # ... and window k+1 is fetched NOW (lock-step host only: HF's per-micro-batch read on), so that the last
        # training_step of window k can enqueue the teacher's pass for its first micro-batch beside the optimizer step
        mode = os.environ.get("SD_WINDOW_AHEAD", "auto")
        if batch_samples and (mode == "1" or (mode == "auto" and self.args.logging_nan_inf_filter)):
            nxt = self._fetch_window(core, epoch_iterator, num_batches, device)
            self._prefetched = (epoch_iterator, nxt)
            self._window_next = list(nxt[0])
            for key, rr in nxt[2]:
                self._rows_ahead[key[0]] = (key, rr, nxt[3])
        return batch_samples, num_items

    def _fetch_window(self, core, epoch_iterator, num_batches, device):
        """HF's fetch of one accumulation window + the loss-row selection of its micro-batches, on the copy stream.
        Returns (batch_samples, num_items_in_batch, [(key, (rows, row_labels))], event)."""
        main = torch.cuda.current_stream(core.flat.device)
        if self._copy_stream is None:
            self._copy_stream = ops.concurrent_stream(core.flat.device, "h2d")
        cs = self._copy_stream
        hip_teacher = isinstance(self.teacher_model, HipQwen3ForCausalLM)
        with torch.cuda.stream(cs):
            batch_samples, num_items = super().get_batch_samples(epoch_iterator, num_batches, device)
            selected = []
            for b in batch_samples:
                lab = b.get("labels") if isinstance(b, dict) else None
                if lab is None or not torch.is_tensor(lab) or not lab.is_cuda:
                    continue
                need_teacher = b.get("teacher_top_k_v") is None and self.teacher_model is not None
                am = b.get("attention_mask")
                tam = b.get("teacher_attention_mask") if hip_teacher and need_teacher else None
                am = am if am is not None and tuple(am.shape) == tuple(lab.shape) else None
                tam = tam if tam is not None and tuple(tam.shape) == tuple(lab.shape) else None
                sm = b.get("speech_token_mask")
                selected.append((self._rows_key(lab, sm, am, tam), ops.loss_rows(lab, sm, right_padded=(am, tam))))
            ready = torch.cuda.Event()
            ready.record(cs)
        # the tensors were allocated on the copy stream and are used on the main and the teacher's stream
        users = [main] + ([self._teacher_stream] if self._teacher_stream is not None else [])
        held = [t for b in batch_samples if isinstance(b, dict) for t in b.values() if torch.is_tensor(t) and t.is_cuda]
        held += [t for _, rr in selected for t in rr] + ([num_items] if torch.is_tensor(num_items) and num_items.is_cuda else [])
        for t in held:
            for st in users:
                t.record_stream(st)
        return batch_samples, num_items, selected, ready

    _TEACHER_SIDE_KEYS = ("teacher_input_ids", "teacher_attention_mask", "speech_token_mask", "teacher_top_k_v", "teacher_top_k_i")

    def training_step(self, model, inputs, *args, **kwargs):
        lab = inputs.get("labels") if isinstance(inputs, dict) else None
        loss = super().training_step(model, inputs, *args, **kwargs)
        mode = os.environ.get("SD_TEACHER_AHEAD", "auto")
        if lab is not None and self._window and (mode == "1" or (mode == "auto" and self.args.logging_nan_inf_filter)):
            self._launch_teacher_ahead(id(lab))
        return loss

    def _launch_teacher_ahead(self, cur_id):
        """Enqueue the frozen teacher's pass for the micro-batch that follows ``cur_id`` in this accumulation window."""
        seq = self._window + self._window_next   # (the next window's first micro-batch follows this window's last)
        idx = next((i for i, b in enumerate(seq) if isinstance(b, dict) and id(b.get("labels")) == cur_id), None)
        if idx is None or idx + 1 >= len(seq):
            return
        nxt = seq[idx + 1]
        lab = nxt.get("labels")
        entry = self._rows_ahead.get(id(lab))
        core = ddp.unwrap(self.model)
        ids, tids = nxt.get("input_ids"), nxt.get("teacher_input_ids")
        if (entry is None or entry[1][0].numel() == 0 or not self.overlap_teacher or nxt.get("teacher_top_k_v") is not None
                or not isinstance(self.teacher_model, HipQwen3ForCausalLM) or not isinstance(core, HipQwen3ForCausalLM)
                or ids is None or (tids is not None and tuple(tids.shape) != tuple(ids.shape))):
            return
        _, (rows, _), fetched = entry
        tam = nxt.get("teacher_attention_mask")
        aligned = tam is not None and tuple(tam.shape) == tuple(lab.shape)
        kw = dict({"padding_checked": True} if aligned or tam is None else {}, concurrent=True)
        if self._teacher_stream is None:
            self._teacher_stream = ops.concurrent_stream(ids.device, "teacher")
        side = self._teacher_stream
        side.wait_event(fetched)
        for t in (ids, tids, tam, nxt.get("attention_mask"), rows):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(side)
        base = {k: v for k, v in nxt.items() if k not in self._TEACHER_SIDE_KEYS}
        with torch.cuda.stream(side):
            self._ahead_results[id(lab)] = self._teacher_pass(base, tids, tam, core.dims.vocab_size, rows, kw)

    def _load_best_model(self):
        """HF's loader looks for ``model.safetensors``; a LoRA student's checkpoints hold the adapter (``_save`` above)."""
        core = ddp.unwrap(self.model)
        best = self.state.best_model_checkpoint
        if isinstance(core, HipQwen3ForCausalLM) and core._lora is not None and best is not None:
            core._lora.load_adapter(best)
            return
        return super()._load_best_model()

    def _teacher_pass(self, inputs, teacher_input_ids, teacher_attention_mask, vocab_size, rows=None, model_kw=None):
        """train.py:60-94: teacher no-grad forward, then on-the-fly top-K unless quantized / top_k <= 0.
        ``rows``: only those flat rows of the logits are produced (our own teacher) or kept (any other module);
        ``model_kw``: extra keyword arguments for a HipQwen3ForCausalLM teacher (``padding_checked``)."""
        model_kw = model_kw or {}
        with torch.no_grad():
            extra = {"logit_rows": rows} if rows is not None and isinstance(self.teacher_model, HipQwen3ForCausalLM) else {}
            if teacher_input_ids is not None:
                teacher_outputs = self.teacher_model(input_ids=teacher_input_ids, attention_mask=teacher_attention_mask,
                                                     **extra, **model_kw)
            else:
                teacher_outputs = self.teacher_model(**{k: v for k, v in inputs.items() if k != "labels"}, **extra,
                                                     **model_kw)
            teacher_logits = teacher_outputs.logits
            if rows is not None and not extra:
                teacher_logits = teacher_logits.reshape(-1, teacher_logits.size(-1))[rows]
            if not self.is_quantized_teacher and self.top_k > 0:
                v, i = self._extract_topk(teacher_logits, self.top_k, vocab_size)
                return None, v, i
        return teacher_logits, None, None

    def compute_loss(self, model, inputs, return_outputs=False, **kwargs):
        teacher_input_ids = inputs.pop("teacher_input_ids", None)
        teacher_attention_mask = inputs.pop("teacher_attention_mask", None)
        speech_mask = inputs.pop("speech_token_mask", None)
        teacher_top_k_v = inputs.pop("teacher_top_k_v", None)
        teacher_top_k_i = inputs.pop("teacher_top_k_i", None)

        need_teacher = teacher_top_k_v is None and self.teacher_model is not None
        ids0 = inputs.get("input_ids")
        if (need_teacher and teacher_input_ids is not None and ids0 is not None
                and tuple(teacher_input_ids.shape) != tuple(ids0.shape)):
            # The collator pads teacher and student separately (data.py:219-278) and the loss indexes both with ONE
            # [B, T-1] row mask (distillation_loss.py:31-45): the reference dies there with an IndexError.  Same
            # condition, named, before any kernel indexes a [B, T_teacher] tensor with the student's T.
            raise ValueError(f"teacher_input_ids {tuple(teacher_input_ids.shape)} and input_ids {tuple(ids0.shape)} must be "
                             "position-aligned (data.py:20-60 align_prefixes): the loss selects rows of both with one mask")
        if model.training and ids0 is not None:
            self.tokens_seen += ids0.numel()
            if self._tps_mark is None:
                self._tps_mark = (time.perf_counter(), self.tokens_seen - ids0.numel())
        # `model` is what the loop trains (HipDataParallel / torch DDP / the bare module); `core` is the module itself
        core = ddp.unwrap(model)
        hip_student = isinstance(core, HipQwen3ForCausalLM)
        hip_teacher = isinstance(self.teacher_model, HipQwen3ForCausalLM)
        vocab = getattr(getattr(core, "dims", None), "vocab_size", None)  # only our own model type is overlapped
        side = None
        fetched = None
        ids = inputs.get("input_ids")
        rows = row_labels = None
        lab = inputs.get("labels")
        checked = {}
        if (self.compact_head and not return_outputs and hip_student and lab is not None
                and lab.is_cuda and isinstance(self.distill_loss_fn, DistillationLoss)):
            # the one host read of the step: the row count, and (same read) that the masks are right-padded
            # only masks laid out on the labels' [B, T] grid go into the fused check; any other keeps its own validation
            # in the model's forward (padding_checked stays unset for it)
            am, tam = inputs.get("attention_mask"), (teacher_attention_mask if hip_teacher and need_teacher else None)
            am = am if am is not None and tuple(am.shape) == tuple(lab.shape) else None
            tam = tam if tam is not None and tuple(tam.shape) == tuple(lab.shape) else None
            key = self._rows_key(lab, speech_mask, am, tam)
            hit = self._rows_ahead.pop(key[0], None)
            if hit is not None and hit[0] == key:   # selected with the accumulation window's other batches (get_batch_samples)
                rows, row_labels = hit[1]
                fetched = hit[2]                    # event: this batch (and its rows) are on the device
            else:
                rows, row_labels = ops.loss_rows(lab, speech_mask, right_padded=(am, tam))
            checked = {"padding_checked": True} if am is not None or inputs.get("attention_mask") is None else {}
            teacher_checked = {"padding_checked": True} if tam is not None or teacher_attention_mask is None else {}
            if rows.numel() == 0:  # N == 0: the full path returns the reference's zeros (distillation_loss.py:47-53)
                rows = row_labels = None
        else:
            teacher_checked = {}
        teacher_kw = teacher_checked if hip_teacher else {}
        ahead = self._ahead_results.pop(id(lab), None) if fetched is not None and rows is not None else None
        if ahead is not None:   # SD_TEACHER_AHEAD: this batch's teacher pass was enqueued at the end of the previous micro-step
            side = self._teacher_stream
            teacher_logits, teacher_top_k_v, teacher_top_k_i = ahead
            if hip_student:
                checked = dict(checked, concurrent=True)
        elif (need_teacher and self.overlap_teacher and vocab is not None and ids is not None and ids.is_cuda
                and hip_teacher):
            if self._teacher_stream is None:
                self._teacher_stream = ops.concurrent_stream(ids.device, "teacher")
            side = self._teacher_stream
            if fetched is not None:   # the teacher needs the batch, not whatever else the main stream still has queued
                side.wait_event(fetched)
                for t in (ids, teacher_input_ids, teacher_attention_mask, inputs.get("attention_mask"), rows):
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(side)
            else:
                side.wait_stream(torch.cuda.current_stream())
            # two passes share the GPU from here to the loss: both are told (SD_FWD_CONCURRENT, include/sd_hip.h)
            teacher_kw = dict(teacher_kw, concurrent=True)
            if hip_student:
                checked = dict(checked, concurrent=True)
            with torch.cuda.stream(side):
                teacher_logits, teacher_top_k_v, teacher_top_k_i = self._teacher_pass(
                    inputs, teacher_input_ids, teacher_attention_mask, vocab, rows, teacher_kw)
        elif rows is not None and teacher_top_k_v is not None:  # pre-extracted top-K: keep the same rows
            dev = lab.device
            teacher_top_k_v = teacher_top_k_v.to(dev).reshape(-1, teacher_top_k_v.size(-1))[rows]
            teacher_top_k_i = teacher_top_k_i.to(dev).reshape(-1, teacher_top_k_i.size(-1))[rows]

        outputs = model(**inputs, **checked) if rows is None else model(**inputs, logit_rows=rows, **checked)  # train.py:54
        student_logits = outputs.logits
        labels = inputs.pop("labels", None)

        teacher_logits_local = None
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
            for t in (teacher_logits, teacher_top_k_v, teacher_top_k_i):   # allocated on the teacher's stream, read on this one
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(torch.cuda.current_stream())
            teacher_logits_local = teacher_logits
        elif need_teacher:  # reference order: student first, then teacher (train.py:60-94)
            teacher_logits_local, teacher_top_k_v, teacher_top_k_i = self._teacher_pass(
                inputs, teacher_input_ids, teacher_attention_mask, student_logits.size(-1), rows, teacher_kw)
        teacher_logits = teacher_logits_local

        if isinstance(self.distill_loss_fn, DistillationLoss):
            self.distill_loss_fn.inplace_grad = (not return_outputs) and hip_student
        if rows is not None:
            loss, task_loss, distill_loss, teacher_loss = self.distill_loss_fn.forward_rows(
                student_logits, row_labels, teacher_logits=teacher_logits, teacher_top_k_v=teacher_top_k_v,
                teacher_top_k_i=teacher_top_k_i)
        else:
            loss, task_loss, distill_loss, teacher_loss = self.distill_loss_fn(
                student_logits=student_logits, labels=labels, teacher_logits=teacher_logits,
                teacher_top_k_v=teacher_top_k_v, teacher_top_k_i=teacher_top_k_i, speech_token_mask=speech_mask)

        if self.state.global_step % self.args.logging_steps == 0:  # train.py:107-114
            # one device->host sync for the three scalars instead of three .item() calls
            vals = torch.stack([task_loss.detach().float(), teacher_loss.detach().float(),
                                distill_loss.detach().float()]).tolist()
            self.log({"student_loss": vals[0], "teacher_loss": vals[1], "distill_loss": vals[2]})
        return (loss, outputs) if return_outputs else loss


def _pretrained_types():
    from transformers import PreTrainedModel
    return (PreTrainedModel,)
