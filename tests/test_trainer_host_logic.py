"""CPU tests of the plug-in's HOST logic (train.py:43-116 control flow) with the compute back-ends
replaced by the oracle.  This is the checker standing in for the kernels in a test; the product
default back-ends are the HIP kernels and refuse CPU tensors (tests/test_cabi.py)."""
import tempfile

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import distill_loss as L


class TinyLM(nn.Module):
    def __init__(self, V=32, h=8):
        super().__init__()
        self.emb = nn.Embedding(V, h)
        self.out = nn.Linear(h, V, bias=False)
        self.calls = 0

    def forward(self, input_ids=None, attention_mask=None, labels=None, **kw):
        self.calls += 1
        return type("O", (dict,), {"logits": property(lambda s: s["logits"])})(logits=self.out(self.emb(input_ids)))


class OracleLoss(nn.Module):
    def __init__(self, T, a):
        super().__init__()
        self.T, self.a, self.seen = T, a, []

    def forward(self, student_logits, labels, teacher_logits=None, teacher_top_k_v=None, teacher_top_k_i=None,
                speech_token_mask=None):
        self.seen.append("dense" if teacher_logits is not None else "sparse")
        return L.distill_loss(student_logits, labels, teacher_logits, teacher_top_k_v, teacher_top_k_i,
                              speech_token_mask, self.T, self.a, acc=torch.float32)


def make(top_k=4, quantized=False, logging_steps=1):
    from transformers import TrainingArguments
    from speech_distill_amd.trainer import DistillationTrainer
    torch.manual_seed(0)
    student, teacher = TinyLM(), TinyLM()
    args = TrainingArguments(output_dir=tempfile.mkdtemp(), use_cpu=True, report_to=[], logging_steps=logging_steps,
                             remove_unused_columns=False, label_names=["labels"], save_strategy="no")
    tr = DistillationTrainer(model=student, args=args, teacher_model=teacher, temperature=2.0, alpha=0.5, top_k=top_k,
                             is_quantized_teacher=quantized)
    tr.distill_loss_fn = OracleLoss(2.0, 0.5)
    tr._extract_topk = lambda logits, k, V: L.extract_topk(logits, k, V)
    logged = []
    tr.log = lambda d, *a, **k: logged.append(d)
    return tr, student, teacher, logged


def batch():
    ids = torch.randint(0, 32, (2, 6), generator=torch.Generator().manual_seed(1))
    labels = ids.clone()
    labels[:, :2] = -100
    return {"input_ids": ids, "attention_mask": torch.ones_like(ids), "labels": labels,
            "teacher_input_ids": ids.clone(), "teacher_attention_mask": torch.ones_like(ids)}


def test_on_the_fly_sparse_path_and_logging():
    tr, student, teacher, logged = make(top_k=4)
    inputs = batch()
    loss = tr.compute_loss(student, inputs)
    assert tr.distill_loss_fn.seen == ["sparse"] and teacher.calls == 1 and student.calls == 1
    assert loss.requires_grad and set(logged[-1]) == {"student_loss", "teacher_loss", "distill_loss"}
    # compute_loss mutates the dict exactly like the reference: the five keys and labels are popped
    assert set(inputs) == {"input_ids", "attention_mask"}


def test_pre_extracted_topk_skips_teacher_forward():
    tr, student, teacher, _ = make(top_k=4)
    b = batch()
    b["teacher_top_k_v"] = -torch.rand(2, 6, 4).half()
    b["teacher_top_k_i"] = torch.randint(0, 32, (2, 6, 4)).int()
    tr.compute_loss(student, b)
    assert teacher.calls == 0 and tr.distill_loss_fn.seen == ["sparse"]


@pytest.mark.parametrize("kw", [dict(top_k=0), dict(quantized=True)])
def test_dense_branch(kw):
    tr, student, teacher, _ = make(**kw)
    tr.compute_loss(student, batch())
    assert tr.distill_loss_fn.seen == ["dense"] and teacher.calls == 1


def test_no_teacher_ids_falls_back_to_student_inputs_and_return_outputs():
    tr, student, teacher, _ = make()
    b = batch()
    del b["teacher_input_ids"], b["teacher_attention_mask"]
    loss, out = tr.compute_loss(student, b, return_outputs=True)
    assert teacher.calls == 1 and out.logits.shape == (2, 6, 32)


def test_logging_cadence_follows_global_step():
    tr, student, _, logged = make(logging_steps=10)
    tr.state.global_step = 3
    tr.compute_loss(student, batch())
    assert not logged
    tr.state.global_step = 20
    tr.compute_loss(student, batch())
    assert len(logged) == 1


def test_matches_reference_value():
    """Same numbers as the oracle step on the same tiny models (control flow does not change the maths)."""
    tr, student, teacher, logged = make(top_k=4)
    b = batch()
    loss = tr.compute_loss(student, dict(b))
    with torch.no_grad():
        s, t = student(input_ids=b["input_ids"]).logits, teacher(input_ids=b["input_ids"]).logits
    v, i = L.extract_topk(t, 4, 32)
    ref = L.distill_loss(s, b["labels"], teacher_top_k_v=v, teacher_top_k_i=i, temperature=2.0, alpha=0.5)
    np.testing.assert_allclose(float(loss), float(ref[0]), rtol=1e-5)


def test_loss_rows_selects_what_the_reference_keeps():
    """ops.loss_rows == the reference's shift + valid mask (distillation_loss.py:31-45): position t < T-1 is kept iff
    labels[t+1] != -100 (and mask[t+1]); the label a kept row predicts is labels[t+1]."""
    from speech_distill_amd import ops
    g = torch.Generator().manual_seed(9)
    B, T = 3, 17
    labels = torch.randint(0, 50, (B, T), generator=g)
    labels[:, :5] = -100
    labels[1, 9:] = -100
    labels[2, 7] = -100
    mask = (torch.rand(B, T, generator=g) > 0.3).long()
    for m in (None, mask):
        rows, lab = ops.loss_rows(labels, m)
        y = labels[:, 1:].reshape(-1)
        valid = y != -100
        if m is not None:
            valid &= m[:, 1:].reshape(-1).bool()
        # flat index into the UNshifted [B, T] grid of the shifted position (b, t)
        want = torch.tensor([b * T + t for b in range(B) for t in range(T - 1)])[valid]
        assert torch.equal(rows, want)
        assert torch.equal(lab, y[valid])
    rows, lab = ops.loss_rows(torch.full((2, 4), -100))
    assert rows.numel() == 0 and lab.numel() == 0


def test_teacher_batch_of_another_length_is_refused_with_both_shapes():
    """The collator pads teacher and student separately (data.py:219-278); the reference then dies in the loss with an
    IndexError (one [B,T-1] mask indexes both, distillation_loss.py:31-45).  Here: a ValueError naming both shapes,
    before anything indexes a [B,T_teacher] tensor with the student's T."""
    tr, student, teacher, _ = make(top_k=4)
    b = batch()
    b["teacher_input_ids"] = torch.cat([b["teacher_input_ids"], b["teacher_input_ids"][:, :2]], 1)
    b["teacher_attention_mask"] = torch.ones_like(b["teacher_input_ids"])
    with pytest.raises(ValueError, match=r"teacher_input_ids \(2, 8\) and input_ids \(2, 6\)"):
        tr.compute_loss(student, b)
    assert teacher.calls == 0 and student.calls == 0
    # pre-extracted top-K: the teacher ids are never used (train.py:58-60), so their length does not matter
    b = batch()
    b["teacher_input_ids"] = torch.cat([b["teacher_input_ids"], b["teacher_input_ids"][:, :2]], 1)
    b["teacher_top_k_v"] = -torch.rand(2, 6, 4).half()
    b["teacher_top_k_i"] = torch.randint(0, 32, (2, 6, 4)).int()
    tr.compute_loss(student, b)


def test_loss_rows_refuses_masks_off_the_label_grid():
    from speech_distill_amd import ops
    labels = torch.randint(0, 50, (2, 6))
    with pytest.raises(ValueError, match=r"attention mask \(2, 8\) != labels \(2, 6\)"):
        ops.loss_rows(labels, right_padded=(torch.ones(2, 8, dtype=torch.long),))
    with pytest.raises(ValueError, match="speech_token_mask"):
        ops.loss_rows(labels, torch.ones(2, 5))


def test_tokens_per_second_is_logged_by_the_real_loop():
    """the counterpart logs tokens/sec.  A real HF ``Trainer.train()`` on the CPU: every training log
    (the dict with ``loss``) carries ``tokens_per_second`` = positions of the micro-batches since the previous log over
    the wall time, and ``tokens_seen`` counts B*T per micro-batch."""
    from transformers import TrainingArguments
    from speech_distill_amd.trainer import DistillationTrainer
    torch.manual_seed(0)
    student, teacher = TinyLM(), TinyLM()
    rows = []
    g = torch.Generator().manual_seed(3)
    for _ in range(16):
        ids = torch.randint(0, 32, (6,), generator=g)
        lab = ids.clone()
        lab[:2] = -100
        rows.append({"input_ids": ids, "labels": lab, "attention_mask": torch.ones(6, dtype=torch.long),
                     "teacher_input_ids": ids.clone(), "teacher_attention_mask": torch.ones(6, dtype=torch.long)})
    args = TrainingArguments(output_dir=tempfile.mkdtemp(), use_cpu=True, report_to=[], logging_steps=1,
                             per_device_train_batch_size=2, gradient_accumulation_steps=2, num_train_epochs=1,
                             remove_unused_columns=False, label_names=["labels"], save_strategy="no",
                             dataloader_num_workers=0)
    tr = DistillationTrainer(model=student, args=args, train_dataset=rows, teacher_model=teacher, top_k=4,
                             data_collator=lambda f: {k: torch.stack([x[k] for x in f]) for k in f[0]})
    tr.distill_loss_fn = OracleLoss(2.0, 0.5)
    tr._extract_topk = lambda logits, k, V: L.extract_topk(logits, k, V)
    tr.train()
    assert tr.tokens_seen == 16 * 6
    train_logs = [d for d in tr.state.log_history if "loss" in d and "learning_rate" in d]
    assert len(train_logs) == 4 and all(d["tokens_per_second"] > 0 for d in train_logs)
    # the sub-loss logs of train.py:107-114 are untouched
    sub = [d for d in tr.state.log_history if "student_loss" in d]
    assert sub and all("tokens_per_second" not in d for d in sub)


def test_rs_ag_fallback_is_loud_and_counted(monkeypatch):
    """algo='rs_ag' on a bucket that does not divide by the world size falls back to all_reduce with ONE
    warning per bucket size and a count in stats -- never silently."""
    import warnings
    from speech_distill_amd import ddp
    red = ddp.FlatGradAllReduce.__new__(ddp.FlatGradAllReduce)
    red.algo, red.world, red.group, red.stats, red._warned_sizes = "rs_ag", 3, None, {}, set()
    calls = []
    monkeypatch.setattr(ddp.dist, "all_reduce", lambda t, op=None, group=None: calls.append(t.numel()))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        red._reduce_on_stream(torch.zeros(8))
        red._reduce_on_stream(torch.zeros(8))
        red._reduce_on_stream(torch.zeros(16))
    assert calls == [8, 8, 16] and red.stats["rs_ag_fallbacks"] == 3
    assert len([x for x in w if "does not divide" in str(x.message)]) == 2  # sizes 8 and 16, once each


def test_rccl_choice_parser_reads_a_debug_log(tmp_path):
    """bench.py's `comm.rccl_choice` from an NCCL_DEBUG=INFO log (format of NCCL/RCCL 2.2x)."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    log = tmp_path / "rccl.log"
    log.write_text(
        "host:1:1 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC\\n"
        "host:1:1 [0] NCCL INFO Channel 01/0 : 0[0] -> 1[1] via P2P/IPC\\n"
        "host:1:1 [0] NCCL INFO Connected all rings\\nhost:1:1 [0] NCCL INFO Connected all trees\\n"
        "host:1:1 [0] NCCL INFO 16 coll channels, 0 collnet channels, 0 nvls channels, 16 p2p channels, 2 p2p channels per peer\\n"
        "host:1:1 [0] NCCL INFO AllReduce: 33030144 Bytes -> Algo 1 proto 2 time 412.5\\n"
        "host:1:1 [0] NCCL INFO AllReduce: 33030144 Bytes -> Algo 1 proto 2 time 412.5\\n"
        "host:1:1 [0] NCCL INFO AllReduce: 4096 Bytes -> Algo 0 proto 0 time 8.1\\n")
    r = bench.rccl_choice(str(log))
    assert r["transports"] == {"P2P/IPC": 2} and r["rings_connected"] and r["trees_connected"]
    assert {"bytes": 33030144, "algo": "ring", "proto": "Simple", "count": 2} in r["collectives"]
    assert {"bytes": 4096, "algo": "tree", "proto": "LL", "count": 1} in r["collectives"]
    assert bench.rccl_choice(None) is None and bench.rccl_choice(str(tmp_path / "absent")) is None
