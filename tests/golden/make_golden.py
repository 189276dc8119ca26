#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Run once in the build container (``python tests/golden/make_golden.py``); the
outputs (*.npz) are committed, the reference's code is not.  It imports

  /root/reference/distillation_loss.py   (DistillationLoss)
  /root/reference/train.py               (DistillationTrainer)       [peft/s3tokenizer/torchaudio stubbed:
  /root/reference/data.py                (ProcessedDataCollator)      they are absent here and unused on this path]
  transformers.models.qwen3              (Qwen3ForCausalLM, random init from a local Qwen3Config)

and records inputs + expected outputs only.  The GPU box never sees
/root/reference; tests read the *.npz files.

Fixtures:
  g1_loss_*.npz      DistillationLoss: (student_logits, teacher | (v,i), labels[, mask], T, alpha)
                     -> (total, task, distill, teacher, d total / d student_logits)
  g2_extract.npz     train.py:80-91 on-the-fly log-softmax + top-K (incl. V_teacher > V_student)
  g3_collator.npz    ProcessedDataCollator on 8 ragged samples
  g4_step_c1.npz     BASELINE config 1: tiny 2-layer teacher + 2-layer student through
                     DistillationTrainer.compute_loss and a short real HF Trainer.train()
  g5_qwen3.npz       HF Qwen3ForCausalLM tiny config: logits + every parameter gradient
Model weights are regenerated from ``oracle.qwen3.init_weights(shape, seed)``; the
fixtures store a checksum of them, not the weights.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import transformers  # noqa: E402
from transformers import (  # noqa: E402,F401  (finish transformers' lazy probes before stubbing)
    AutoModelForCausalLM, AutoTokenizer, BitsAndBytesConfig, Trainer, TrainingArguments)
from transformers import Qwen3Config, Qwen3ForCausalLM  # noqa: E402

for name in ("s3tokenizer", "torchaudio", "peft"):
    if name not in sys.modules:
        m = types.ModuleType(name)
        if name == "peft":
            m.LoraConfig = object
            m.get_peft_model = lambda *a, **k: None
        sys.modules[name] = m
sys.path.insert(0, "/root/reference")
import data as ref_data  # noqa: E402
import train as ref_train  # noqa: E402
from distillation_loss import DistillationLoss as RefLoss  # noqa: E402

from oracle import qwen3 as OQ  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(4)


def checksum(w):
    """Order-stable fp64 checksum of a weight dict."""
    return np.array([float(v.double().sum()) + float((v.double() ** 2).sum()) for _, v in sorted(w.items())])


def hf_model(shape: OQ.Qwen3Shape, w, attn="eager"):
    cfg = Qwen3Config(
        vocab_size=shape.vocab_size, hidden_size=shape.hidden_size, intermediate_size=shape.intermediate_size,
        num_hidden_layers=shape.num_hidden_layers, num_attention_heads=shape.num_attention_heads,
        num_key_value_heads=shape.num_key_value_heads, head_dim=shape.head_dim, rms_norm_eps=shape.rms_norm_eps,
        rope_theta=shape.rope_theta, tie_word_embeddings=shape.tie_word_embeddings, attention_bias=False,
        max_position_embeddings=4096, attn_implementation=attn, use_cache=False)
    model = Qwen3ForCausalLM(cfg).float()
    sd = {k: v.clone() for k, v in w.items()}
    if shape.tie_word_embeddings:
        sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("rotary" in m or "lm_head" in m for m in missing), missing
    if shape.tie_word_embeddings:
        model.tie_weights()
        assert model.lm_head.weight.data_ptr() == model.model.embed_tokens.weight.data_ptr()
    return model


# ----------------------------------------------------------------------------- G1
def g1():
    cases = []
    g = torch.Generator().manual_seed(11)

    def mk(B, T, V, K, dtype, prefix, mask=False, empty=False, label_miss=False, name=""):
        s = torch.randn(B, T, V, generator=g) * 3.0
        t = torch.randn(B, T, V, generator=g) * 3.0
        if dtype == "bf16":
            s, t = s.bfloat16(), t.bfloat16()
        labels = torch.randint(0, V, (B, T), generator=g)
        labels[:, :prefix] = -100
        if B > 1:
            labels[1, T - 3:] = -100  # right padding on one row
        if empty:
            labels[:] = -100
        sm = None
        if mask:
            sm = (torch.rand(B, T, generator=g) > 0.3).float()
        # sparse inputs exactly as train.py:85-91 makes them
        lp = torch.log_softmax(t.float(), dim=-1)
        v, i = torch.topk(lp, K, dim=-1)
        v, i = v.half(), i.int()
        if label_miss:
            # make sure some labels are outside the teacher's top-K and some inside
            for b in range(B):
                for tt in range(prefix, T - 1, 2):
                    labels[b, tt + 1] = int(i[b, tt, 0]) if (tt // 2) % 2 == 0 else int(lp[b, tt].argmin())
        out = {}
        for mode in ("dense", "sparse"):
            for (Tm, al) in ((2.0, 0.5), (1.0, 0.3), (4.0, 0.0)):
                sl = s.clone().requires_grad_(True)
                fn = RefLoss(temperature=Tm, alpha=al)
                kw = dict(teacher_logits=t) if mode == "dense" else dict(teacher_top_k_v=v, teacher_top_k_i=i)
                res = fn(sl, labels, speech_token_mask=sm, **kw)
                if res[0].requires_grad:
                    res[0].backward()
                    grad = sl.grad.float().numpy()
                else:
                    grad = np.zeros(s.shape, np.float32)
                key = f"{mode}_T{Tm}_a{al}"
                out[key + "_losses"] = np.array([float(r) for r in res], np.float64)
                out[key + "_grad"] = grad
        out.update(student=s.float().numpy(), teacher=t.float().numpy(), labels=labels.numpy(),
                   top_v=v.numpy(), top_i=i.numpy(), dtype=np.array(dtype), K=np.array(K))
        if sm is not None:
            out["speech_mask"] = sm.numpy()
        np.savez_compressed(os.path.join(HERE, f"g1_loss_{name}.npz"), **out)
        cases.append(name)

    mk(2, 16, 64, 8, "fp32", 4, name="small_fp32")
    mk(2, 16, 64, 8, "bf16", 4, name="small_bf16")
    mk(2, 33, 1000, 100, "fp32", 9, label_miss=True, name="v1000_k100_fp32")
    mk(2, 33, 1000, 128, "bf16", 9, name="v1000_k128_bf16")
    mk(2, 16, 64, 8, "fp32", 4, mask=True, name="small_mask_fp32")
    mk(2, 16, 64, 8, "fp32", 4, empty=True, name="small_empty_fp32")
    print("G1", cases)


# ----------------------------------------------------------------------------- G2
def g2():
    g = torch.Generator().manual_seed(22)
    out = {}
    for name, (B, T, Vt, Vs, K) in {"a": (2, 9, 1000, 1000, 100), "b": (2, 9, 1100, 1000, 128),
                                    "c": (1, 5, 4096, 4096, 16)}.items():
        t = torch.randn(B, T, Vt, generator=g) * 4.0
        tr = t[..., :Vs]  # train.py:82-83
        lp = torch.nn.functional.log_softmax(tr, dim=-1)  # train.py:85
        v, i = torch.topk(lp, k=K, dim=-1)  # train.py:86-88
        out[f"{name}_logits"] = t.numpy()
        out[f"{name}_v"] = v.to(torch.float16).numpy()  # train.py:90
        out[f"{name}_i"] = i.to(torch.int32).numpy()  # train.py:91
        out[f"{name}_meta"] = np.array([Vs, K])
    np.savez_compressed(os.path.join(HERE, "g2_extract.npz"), **out)
    print("G2 ok")


# ----------------------------------------------------------------------------- G3
class DuckTok:
    """Duck-typed tokenizer: the collator only needs pad_token(_id) and encode()."""
    pad_token = "<|semantic_token_end|>"

    def __init__(self, pad_id, bos_id):
        self.pad_token_id, self.bos_id = pad_id, bos_id

    def encode(self, text, add_special_tokens=False):
        assert text == "<|semantic_token_start|>"
        return [self.bos_id]


def ragged_samples(n, V, text_lo, bos_id, pad_id, speech_lo, g, max_len, K=None, no_bos_row=None):
    feats = []
    for r in range(n):
        L = int(torch.randint(max_len // 2, max_len + 1, (1,), generator=g))
        n_text = int(torch.randint(3, max(4, L // 3), (1,), generator=g))
        text = torch.randint(0, text_lo, (n_text,), generator=g)
        speech = torch.randint(speech_lo, V, (L - n_text - 2,), generator=g)
        speech[speech == pad_id] = speech_lo
        speech[speech == bos_id] = speech_lo
        ids = torch.cat([text, torch.tensor([bos_id]), speech, torch.tensor([pad_id])])  # ends with the EOS==pad (Q2)
        if no_bos_row is not None and r == no_bos_row:
            ids[ids == bos_id] = 1
        f = {"student_input_ids": ids.tolist(), "student_attention_mask": [1] * len(ids),
             "teacher_input_ids": ids.tolist(), "teacher_attention_mask": [1] * len(ids)}
        if K:
            f["teacher_top_k_v"] = (-torch.rand(len(ids), K, generator=g) * 5).half().numpy()
            f["teacher_top_k_i"] = torch.randint(0, V, (len(ids), K), generator=g).int().numpy()
        feats.append(f)
    return feats


def g3():
    g = torch.Generator().manual_seed(33)
    V, bos, pad = 640, 500, 501
    feats = ragged_samples(8, V, 400, bos, pad, 512, g, 40, K=4, no_bos_row=5)
    coll = ref_data.ProcessedDataCollator(DuckTok(pad, bos), pad_token_id=pad)
    batch = coll([dict(f) for f in feats])
    out = {f"out_{k}": v.numpy() for k, v in batch.items()}
    out["n"] = np.array(len(feats))
    for r, f in enumerate(feats):
        out[f"in_{r}_ids"] = np.array(f["student_input_ids"])
        out[f"in_{r}_v"] = f["teacher_top_k_v"]
        out[f"in_{r}_i"] = f["teacher_top_k_i"]
    out["meta"] = np.array([V, bos, pad])
    np.savez_compressed(os.path.join(HERE, "g3_collator.npz"), **out)
    print("G3", {k: tuple(v.shape) for k, v in batch.items()})


# ----------------------------------------------------------------------------- G4 / G5
C1_STUDENT = OQ.Qwen3Shape(640, 128, 256, 2, 2, 1)
C1_TEACHER = OQ.Qwen3Shape(640, 256, 512, 2, 4, 2)


def g5():
    shape = OQ.Qwen3Shape(520, 128, 192, 2, 4, 2)
    w = OQ.init_weights(shape, seed=5, norm_jitter=0.1)
    model = hf_model(shape, w)
    g = torch.Generator().manual_seed(55)
    ids = torch.randint(0, shape.vocab_size, (2, 24), generator=g)
    am = torch.ones_like(ids)
    am[1, 19:] = 0  # right padding on row 1
    logits = model(input_ids=ids, attention_mask=am).logits
    probe = torch.randn(logits.shape, generator=g)
    (logits * probe * am[..., None]).sum().backward()
    out = {"ids": ids.numpy(), "am": am.numpy(), "logits": logits.detach().numpy(), "probe": probe.numpy(),
           "wsum": checksum(w), "shape": np.array([520, 128, 192, 2, 4, 2])}
    for k, p in model.named_parameters():
        if k == "lm_head.weight":
            continue
        out["gnorm_" + k] = np.array(float(p.grad.double().norm()))
        if "layers.1." not in k:  # full gradients for embed, layer 0 and the final norm; norms for the rest
            out["grad_" + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "g5_qwen3.npz"), **out)
    print("G5 logits", tuple(logits.shape))


def g4():
    g = torch.Generator().manual_seed(44)
    V, bos, pad = 640, 500, 501
    sw = OQ.init_weights(C1_STUDENT, seed=1)
    tw = OQ.init_weights(C1_TEACHER, seed=2)
    feats = ragged_samples(8, V, 400, bos, pad, 512, g, 128)
    for f in feats[:2]:  # make sure the longest rows are exactly seq_len 128
        pass
    tok = DuckTok(pad, bos)
    coll = ref_data.ProcessedDataCollator(tok, pad_token_id=pad)
    out = {"meta": np.array([V, bos, pad]), "sw_sum": checksum(sw), "tw_sum": checksum(tw), "n": np.array(8)}
    for r, f in enumerate(feats):
        out[f"in_{r}_ids"] = np.array(f["student_input_ids"])

    tmp = tempfile.mkdtemp()

    def make_trainer(top_k, steps=1, lr=1e-3):
        student = hf_model(C1_STUDENT, sw)
        teacher = hf_model(C1_TEACHER, tw)
        teacher.eval()
        for p in teacher.parameters():
            p.requires_grad_(False)
        args = TrainingArguments(
            output_dir=tmp, per_device_train_batch_size=4, gradient_accumulation_steps=2, num_train_epochs=steps,
            learning_rate=lr, logging_steps=1, save_strategy="no", eval_strategy="no", report_to=[], use_cpu=True,
            remove_unused_columns=False, label_names=["labels"], seed=42, data_seed=42, lr_scheduler_type="constant",
            warmup_steps=0, weight_decay=0.0, max_grad_norm=1.0, dataloader_num_workers=0)

        class DS(torch.utils.data.Dataset):
            def __len__(self):
                return len(feats)

            def __getitem__(self, i):
                return dict(feats[i])

        tr = ref_train.DistillationTrainer(
            model=student, args=args, train_dataset=DS(), data_collator=coll, teacher_model=teacher,
            temperature=2.0, alpha=0.5, top_k=top_k)
        return tr, student

    # (a) compute_loss on the two fixed micro-batches, sparse (top_k=16) and dense (top_k=0)
    for mode, top_k in (("sparse", 16), ("dense", 0)):
        tr, student = make_trainer(top_k)
        logged = []
        tr.log = lambda d, *a, **k: logged.append(dict(d))
        for mb in range(2):
            batch = coll([dict(f) for f in feats[4 * mb: 4 * mb + 4]])
            student.zero_grad()
            loss = tr.compute_loss(student, dict(batch))
            loss.backward()
            out[f"{mode}_mb{mb}_loss"] = np.array(float(loss))
            out[f"{mode}_mb{mb}_logged"] = np.array([logged[-1]["student_loss"], logged[-1]["teacher_loss"],
                                                     logged[-1]["distill_loss"]])
            for k, p in student.named_parameters():
                if k == "lm_head.weight":
                    continue
                out[f"{mode}_mb{mb}_gnorm_{k}"] = np.array(float(p.grad.double().norm()))
            out[f"{mode}_mb{mb}_grad_embed"] = student.model.embed_tokens.weight.grad.numpy().copy()
            out[f"{mode}_mb{mb}_grad_l0_q"] = student.model.layers[0].self_attn.q_proj.weight.grad.numpy().copy()
            out[f"{mode}_mb{mb}_grad_l1_down"] = student.model.layers[1].mlp.down_proj.weight.grad.numpy().copy()

    # (b) short real Trainer.train(): 3 epochs x (8 samples / (4 x GA 2)) = 3 optimizer steps, sequential sampler
    tr, student = make_trainer(16, steps=3)
    tr._get_train_sampler = lambda *a, **k: torch.utils.data.SequentialSampler(tr.train_dataset)
    tr.train()
    hist = [h for h in tr.state.log_history if "loss" in h]
    out["train_loss_per_step"] = np.array([h["loss"] for h in hist])
    out["train_grad_norm_per_step"] = np.array([h.get("grad_norm", np.nan) for h in hist])
    sub = [h for h in tr.state.log_history if "student_loss" in h]
    out["train_sublosses"] = np.array([[h["student_loss"], h["teacher_loss"], h["distill_loss"]] for h in sub])
    out["train_final_wsum"] = checksum({k: v.detach() for k, v in student.state_dict().items()
                                        if k != "lm_head.weight"})
    np.savez_compressed(os.path.join(HERE, "g4_step_c1.npz"), **out)
    print("G4 train loss", out["train_loss_per_step"], "sub", out["train_sublosses"].shape)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g5", "g4"]
    for w_ in which:
        globals()[w_]()
    print("transformers", transformers.__version__, "torch", torch.__version__)
