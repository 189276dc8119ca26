#!/usr/bin/env python3
"""Stage-2 distillation loop on the MI355X path: thin counterpart of the reference's ``train.py``
(train.py:119-420 orchestration, :430-596 flags).  Same flag names where they exist in the reference;
additions are marked (+).  No hub access is assumed: models come from local HF directories, or
``--random_init`` builds the real shapes (or ``--tiny``) with HF default init.

    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train.py --dataset_path ... \
        --teacher_model /models/SoulX-Podcast-1.7B --student_model /models/qwen3-0.6b-expanded
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--teacher_model", default=None)           # train.py:432-437
    p.add_argument("--student_model", default=None)           # train.py:438-443
    p.add_argument("--dataset_path", default=None)            # train.py:444-449 (load_from_disk directory)
    p.add_argument("--output_dir", default="./student_distilled")
    p.add_argument("--max_length", type=int, default=512)     # train.py:454-456
    p.add_argument("--epochs", type=int, default=3)
    p.add_argument("--learning_rate", type=float, default=5e-5)
    p.add_argument("--temperature", type=float, default=2.0)  # train.py:488-491
    p.add_argument("--alpha", type=float, default=0.5)        # train.py:492-496
    p.add_argument("--top_k", type=int, default=128)          # train.py:579-583
    p.add_argument("--per_device_train_batch_size", type=int, default=4)   # (+) hard-coded 4 at train.py:333
    p.add_argument("--gradient_accumulation_steps", type=int, default=4)   # (+) hard-coded 4 at train.py:336
    p.add_argument("--logging_steps", type=int, default=10)                # (+) hard-coded at train.py:338
    p.add_argument("--pad_token_id", type=int, default=153478)
    p.add_argument("--speech_bos_id", type=int, default=None, help="(+) id of <|semantic_token_start|> when no tokenizer dir")
    p.add_argument("--random_init", action="store_true", help="(+) build teacher/student shapes without weights")
    p.add_argument("--tiny", action="store_true", help="(+) BASELINE config-1-sized models (plumbing runs)")
    p.add_argument("--synthetic_samples", type=int, default=0, help="(+) generate N synthetic pre-processed samples")
    p.add_argument("--max_steps", type=int, default=-1)
    return p.parse_args()


class _BosTok:
    """Stand-in tokenizer when only ids are known (the collator needs pad_token(_id) and encode(speech_bos))."""
    pad_token = "<|semantic_token_end|>"

    def __init__(self, pad, bos):
        self.pad_token_id, self.bos = pad, bos

    def encode(self, text, add_special_tokens=False):
        return [self.bos]


def build_models(cfg, dev):
    import speech_distill_amd as sda
    if cfg.random_init:
        sd, td = ((sda.Qwen3Dims(640, 128, 256, 2, 2, 1), sda.Qwen3Dims(640, 256, 512, 2, 4, 2)) if cfg.tiny else
                  (sda.Qwen3Dims.student_06b(), sda.Qwen3Dims.teacher_17b()))
        return sda.HipQwen3ForCausalLM(sd, device=dev, seed=0), sda.HipQwen3ForCausalLM(td, device=dev, seed=1)
    from transformers import AutoModelForCausalLM
    out = []
    for path in (cfg.student_model, cfg.teacher_model):
        hf = AutoModelForCausalLM.from_pretrained(path, torch_dtype=torch.bfloat16)  # train.py:155-178, CPU load
        c = hf.config
        m = sda.HipQwen3ForCausalLM(
            sda.Qwen3Dims(c.vocab_size, c.hidden_size, c.intermediate_size, c.num_hidden_layers, c.num_attention_heads,
                          c.num_key_value_heads, getattr(c, "head_dim", 128), c.rms_norm_eps,
                          getattr(c, "rope_theta", 1e6), c.tie_word_embeddings), device=dev, config=c, init_std=0)
        m.load_hf_state_dict(hf.state_dict())
        out.append(m)
        del hf
    return out[0], out[1]


def main():
    cfg = parse_args()
    from transformers import TrainingArguments
    import speech_distill_amd as sda
    from speech_distill_amd import ddp
    from speech_distill_amd.collator import ProcessedDataCollator
    from speech_distill_amd.trainer import DistillationTrainer
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    student, teacher = build_models(cfg, dev)
    teacher.eval().requires_grad_(False)               # train.py:165-169
    student.gradient_checkpointing_enable()            # train.py:204-208 (accepted; activations are kept in HBM)
    V = student.dims.vocab_size
    bos = cfg.speech_bos_id if cfg.speech_bos_id is not None else min(152927, V - 2)
    pad = cfg.pad_token_id if cfg.pad_token_id < V else V - 1
    if cfg.synthetic_samples:
        g = torch.Generator().manual_seed(1234 + int(os.environ.get("RANK", 0)))
        rows = []
        for _ in range(cfg.synthetic_samples):
            n = int(torch.randint(cfg.max_length // 2, cfg.max_length + 1, (1,), generator=g))
            nt = max(2, n // 4)
            ids = torch.cat([torch.randint(0, min(bos, V), (nt,), generator=g), torch.tensor([bos]),
                             torch.randint(bos + 1, V, (n - nt - 2,), generator=g), torch.tensor([pad])])
            ids[nt + 1:-1][ids[nt + 1:-1] == pad] = bos + 1
            rows.append({"student_input_ids": ids.tolist(), "student_attention_mask": [1] * n,
                         "teacher_input_ids": ids.tolist(), "teacher_attention_mask": [1] * n})
        dataset = rows
    else:
        from datasets import load_from_disk
        dataset = load_from_disk(cfg.dataset_path)     # train.py:234-236 (pre-processed columns, data.py:124-141)
    args = TrainingArguments(
        output_dir=cfg.output_dir, per_device_train_batch_size=cfg.per_device_train_batch_size,
        gradient_accumulation_steps=cfg.gradient_accumulation_steps, num_train_epochs=cfg.epochs,
        learning_rate=cfg.learning_rate, logging_steps=cfg.logging_steps, bf16=True, save_strategy="no",
        eval_strategy="no", report_to=[], remove_unused_columns=False, label_names=["labels"], max_steps=cfg.max_steps,
        dataloader_num_workers=0)                      # train.py:331-354
    trainer = DistillationTrainer(model=student, args=args, train_dataset=dataset,
                                  data_collator=ProcessedDataCollator(_BosTok(pad, bos), pad_token_id=pad),
                                  teacher_model=teacher, temperature=cfg.temperature, alpha=cfg.alpha, top_k=cfg.top_k)
    if int(os.environ.get("WORLD_SIZE", 1)) > 1:
        ddp.attach(student)
    t0 = time.time()
    trainer.train()                                    # train.py:420
    if trainer.is_world_process_zero():
        print(f"done in {time.time() - t0:.1f}s; log tail: {trainer.state.log_history[-3:]}")


if __name__ == "__main__":
    main()
