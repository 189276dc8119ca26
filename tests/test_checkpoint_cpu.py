"""Checkpointing through the real HF Trainer (SURVEY section 5 "Checkpoint / resume", section 8b model protocol).

The reference trains with save_strategy="epoch", load_best_model_at_end=True, save_total_limit=3
(train.py:341-345): HF Trainer calls ``save_model`` on the student at every epoch end and ``load_state_dict`` at the
end.  The HIP student keeps every parameter as a view of one flat buffer with a tied lm_head; these tests build it on
the CPU (construction, state dict and file I/O need no kernel) and check that the checkpoint directory is what an HF
``Qwen3ForCausalLM.save_pretrained`` would have written: loadable by HF itself with identical tensors."""
import os
import tempfile

import pytest
import torch

import speech_distill_amd as sda
from speech_distill_amd.ops import left_padded


def _tiny(tied=True, seed=0):
    return sda.HipQwen3ForCausalLM(sda.Qwen3Dims(64, 128, 64, 2, 2, 1, tie_word_embeddings=tied), device="cpu", seed=seed)


def _trainer(model, out):
    from transformers import TrainingArguments
    from speech_distill_amd.trainer import DistillationTrainer
    args = TrainingArguments(output_dir=out, use_cpu=True, report_to=[], save_strategy="no", remove_unused_columns=False,
                             label_names=["labels"])
    return DistillationTrainer(model=model, args=args, teacher_model=None)


@pytest.mark.parametrize("tied", [True, False])
def test_trainer_save_model_is_an_hf_checkpoint(tied):
    from transformers import Qwen3ForCausalLM
    m = _tiny(tied)
    out = tempfile.mkdtemp()
    tr = _trainer(m, out)
    assert m._params["model.embed_tokens.weight"].data_ptr() == m.flat.data_ptr()  # Trainer's model.to() kept the views
    d = os.path.join(out, "ckpt")
    tr.save_model(d)  # raised "Some tensors share memory" (tied views of the flat buffer)
    assert {"config.json", "model.safetensors", "training_args.bin"} <= set(os.listdir(d))
    hf = Qwen3ForCausalLM.from_pretrained(d, dtype=torch.bfloat16)
    hsd = hf.state_dict()
    ours = m.state_dict()
    assert ("lm_head.weight" in ours) == (not tied)
    for k, v in ours.items():
        assert torch.equal(hsd[k], v), k
    assert torch.equal(hsd["lm_head.weight"], m.lm_head.weight)
    assert hf.config.tie_word_embeddings == tied and hf.config.num_hidden_layers == 2
    # and back: into a fresh HIP-layout model, from the directory and from HF's own state dict (has lm_head.weight)
    m2 = sda.HipQwen3ForCausalLM.from_pretrained(d, device="cpu")
    assert torch.equal(m2.flat, m.flat) and m2.dims == m.dims
    m3 = _tiny(tied, seed=5)
    res = m3.load_state_dict(hsd)
    assert not res.missing_keys and not res.unexpected_keys and torch.equal(m3.flat, m.flat)


def test_load_best_model_at_end_path():
    """HF trainer.py `_load_best_model`: safetensors.load_file -> model.load_state_dict(state_dict, False)."""
    m = _tiny()
    out = tempfile.mkdtemp()
    tr = _trainer(m, out)
    d = os.path.join(out, "checkpoint-1")
    tr.save_model(d)
    saved = m.flat.clone()
    with torch.no_grad():
        m.flat.add_(1.0)
    tr.state.best_model_checkpoint = d
    tr._load_best_model()
    assert torch.equal(m.flat, saved)
    # the parameters are still views of the flat buffer (load copies in place)
    assert m._params["model.layers.1.mlp.down_proj.weight"].data_ptr() > m.flat.data_ptr()
    # strict loading reports what torch would
    sd = m.state_dict()
    sd.pop("model.norm.weight")
    sd["bogus"] = torch.zeros(1)
    with pytest.raises(RuntimeError):
        m.load_state_dict(sd)
    res = m.load_state_dict(sd, strict=False)
    assert res.missing_keys == ["model.norm.weight"] and res.unexpected_keys == ["bogus"]
    sd = m.state_dict()
    sd["model.norm.weight"] = torch.zeros(3)
    with pytest.raises(RuntimeError, match="size mismatch"):
        m.load_state_dict(sd)


def test_flat_adamw_state_dict_round_trip():
    from speech_distill_amd.optim import FlatAdamW
    m = _tiny()
    opt = FlatAdamW(m, lr=1e-3)
    opt.exp_avg.fill_(0.5)
    opt.exp_avg_sq.fill_(0.25)
    opt._step = 7
    f = os.path.join(tempfile.mkdtemp(), "optimizer.pt")
    torch.save(opt.state_dict(), f)
    opt2 = FlatAdamW(_tiny(), lr=5e-5)
    opt2.load_state_dict(torch.load(f, weights_only=False))
    assert opt2._step == 7 and torch.equal(opt2.exp_avg, opt.exp_avg) and torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq)
    assert opt2.param_groups[0]["lr"] == 1e-3


def test_only_bf16_and_right_padding():
    m = _tiny()
    with pytest.raises(TypeError):
        m.float()
    assert not bool(left_padded(torch.tensor([[1, 1, 0, 0], [1, 1, 1, 1], [0, 0, 0, 0]])))
    assert bool(left_padded(torch.tensor([[1, 1, 1, 1], [0, 0, 1, 1]])))
    assert bool(left_padded(torch.tensor([[1, 0, 1, 0]])))
    from speech_distill_amd import ops
    labels = torch.tensor([[-100, 3, 4, 5]])
    ops.loss_rows(labels, None, right_padded=(torch.tensor([[1, 1, 1, 0]]), None))
    with pytest.raises(ValueError, match="right-padded"):
        ops.loss_rows(labels, None, right_padded=(torch.tensor([[0, 1, 1, 1]]),))
