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
    p.add_argument("--num_train_epochs", "--epochs", dest="epochs", type=int, default=3)      # train.py:501-503
    p.add_argument("--warmup_steps", type=int, default=1000)                                   # train.py:504-506
    p.add_argument("--learning_rate", type=float, default=5e-5)
    p.add_argument("--temperature", type=float, default=2.0)  # train.py:488-491
    p.add_argument("--alpha", type=float, default=0.5)        # train.py:492-496
    p.add_argument("--top_k", type=int, default=128)          # train.py:579-583
    p.add_argument("--per_device_train_batch_size", type=int, default=4)   # (+) hard-coded 4 at train.py:333
    p.add_argument("--gradient_accumulation_steps", type=int, default=4)   # (+) hard-coded 4 at train.py:336
    p.add_argument("--logging_steps", type=int, default=10)                # (+) hard-coded at train.py:338
    # train.py:507-517: bf16 and gradient checkpointing are on by default in the reference and cannot be switched off
    # from its command line either (store_true + set_defaults(True)); this path computes in bf16 only
    p.add_argument("--bf16", action="store_true", default=True)
    p.add_argument("--gradient_checkpointing", action="store_true", default=True)
    p.add_argument("--test_size", "--eval_samples", dest="eval_samples", type=int, default=10,
                   help="held-out samples for the per-epoch evaluation (train.py:518-523, 262-269)")
    p.add_argument("--report_to", default="none", help="train.py:524-529 (default there: wandb; no network here)")
    # train.py:530-535 (default there: 1, for on-the-fly tokenisation).  Pre-processed batches collate in microseconds
    # (vectorised collator), so the default here is the main process; workers are forked AFTER the GPU is initialised
    p.add_argument("--dataloader_num_workers", type=int, default=0)
    p.add_argument("--dataloader_prefetch_factor", type=int, default=2)  # train.py:536-541
    p.add_argument("--blocking_h2d", action="store_true", help="(+) accelerate's default blocking host-to-device copies of "
                   "the batch tensors (default here: non_blocking from the pinned buffers of the dataloader)")
    p.add_argument("--logging_nan_inf_filter", type=lambda v: v.lower() not in ("0", "false", "no"), default=True,
                   help="(+) HF TrainingArguments.logging_nan_inf_filter (HF default True = what the reference runs with): "
                        "False drops HF's isnan/isinf host read of the loss after every micro-batch")
    # token strings (train.py:542-577): resolved through the tokenizer found in --student_model; without a tokenizer
    # directory the two ids the pre-processed path needs are given directly (--pad_token_id / --speech_bos_id)
    p.add_argument("--speech_bos", default="<|semantic_token_start|>")
    p.add_argument("--speech_eos", default="<|semantic_token_end|>")
    p.add_argument("--pad_token", default="<|semantic_token_end|>")
    p.add_argument("--text_bos", default="<|text_start|>")
    p.add_argument("--text_eos", default="<|text_end|>")
    p.add_argument("--text_prefix", default='{"en": "", "zh": "", "yue": "<|Yue|>"}')
    p.add_argument("--teacher_prefix", default="<|task_podcast|><|SPEAKER_0|>")
    p.add_argument("--student_prefix", default="")
    p.add_argument("--pad_token_id", type=int, default=153478, help="(+) when --student_model holds no tokenizer")
    p.add_argument("--speech_bos_id", type=int, default=None, help="(+) id of <|semantic_token_start|> when no tokenizer dir")
    # train.py:585-594: a quantised teacher switches the trainer to DENSE distillation (train.py:74-79).  Here the
    # teacher always stays bf16 in HBM (3.5 GB of 288); the flags keep their effect on the loss
    p.add_argument("--load_teacher_in_4bit", action="store_true")
    p.add_argument("--load_teacher_in_8bit", action="store_true")
    # LoRA (train.py:180-202, :470-487): speech_distill_amd/lora.py, merged-weight form
    p.add_argument("--use_lora", action="store_true")
    p.add_argument("--lora_r", type=int, default=32)
    p.add_argument("--lora_alpha", type=int, default=64)
    p.add_argument("--use_rslora", action="store_true", default=True)
    p.add_argument("--init_lora_weights", default="pissa")
    p.add_argument("--random_init", action="store_true", help="(+) build teacher/student shapes without weights")
    p.add_argument("--tiny", action="store_true", help="(+) BASELINE config-1-sized models (plumbing runs)")
    p.add_argument("--synthetic_samples", type=int, default=0, help="(+) generate N synthetic pre-processed samples")
    p.add_argument("--max_steps", type=int, default=-1)
    p.add_argument("--save_strategy", default="epoch", help="(+) train.py:341 hard-codes \"epoch\" (with eval per epoch, "
                   "load_best_model_at_end, save_total_limit=3); \"no\" turns checkpoints and evaluation off")
    p.add_argument("--ddp_backend", default=None, help="(+) torch.distributed backend (default: nccl = RCCL on ROCm)")
    p.add_argument("--equal_length", action="store_true", help="(+) synthetic samples all max_length long (equal N per rank)")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--recompute", default="auto", choices=["auto", "always", "never"],
                   help="(+) what gradient checkpointing (on, as in train.py:512-517) does: layer recompute always, never, "
                        "or only when the activations would take more than a quarter of HBM")
    p.add_argument("--log_json", default=None, help="(+) write trainer.state.log_history there (rank 0; every rank if the "
                   "path contains {rank})")
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
    out = []
    for path in (cfg.student_model, cfg.teacher_model):     # train.py:155-178 (local HF checkpoint directories)
        out.append(sda.HipQwen3ForCausalLM.from_pretrained(path, device=dev))
    return out[0], out[1]


def main():
    cfg = parse_args()
    from transformers import TrainingArguments
    import speech_distill_amd as sda
    from speech_distill_amd.collator import ProcessedDataCollator
    from speech_distill_amd.trainer import DistillationTrainer
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    student, teacher = build_models(cfg, dev)
    teacher.eval().requires_grad_(False)               # train.py:165-169
    if cfg.use_lora:                                   # train.py:180-203
        from speech_distill_amd import lora
        print("Applying LoRA to student model...")
        init = cfg.init_lora_weights
        init = True if init in ("default", "true", "True") else init
        student = lora.get_lora_model(student, lora.LoraConfig(
            r=cfg.lora_r, lora_alpha=cfg.lora_alpha, use_rslora=cfg.use_rslora, init_lora_weights=init))
        lora.print_trainable_parameters(student)
    # train.py:204-208; layer-granular recompute only when the policy asks for it ("auto": when HBM would run short)
    student.gradient_checkpointing_enable(gradient_checkpointing_kwargs={"recompute": cfg.recompute})
    V = student.dims.vocab_size
    bos = cfg.speech_bos_id if cfg.speech_bos_id is not None else (152927 if V > 152928 else V // 2)
    pad = cfg.pad_token_id if cfg.pad_token_id < V else V - 1
    tokenizer = _BosTok(pad, bos)
    if cfg.student_model and any(os.path.exists(os.path.join(cfg.student_model, f))
                                 for f in ("tokenizer.json", "tokenizer_config.json", "vocab.json")):
        from transformers import AutoTokenizer             # train.py:210-231: tokenizer of the student, pad token set
        tokenizer = AutoTokenizer.from_pretrained(cfg.student_model)
        tokenizer.pad_token = cfg.pad_token
        pad, bos = tokenizer.pad_token_id, tokenizer.encode(cfg.speech_bos, add_special_tokens=False)[0]
    if cfg.synthetic_samples:
        g = torch.Generator().manual_seed(1234)        # the same dataset on every rank; the sampler shards it
        rows = []
        for _ in range(cfg.synthetic_samples):
            n = cfg.max_length if cfg.equal_length else int(torch.randint(cfg.max_length // 2, cfg.max_length + 1, (1,), generator=g))
            nt = max(2, n // 4)
            ids = torch.cat([torch.randint(0, min(bos, V), (nt,), generator=g), torch.tensor([bos]),
                             torch.randint(bos + 1, V, (n - nt - 2,), generator=g), torch.tensor([pad])])
            ids[nt + 1:-1][ids[nt + 1:-1] == pad] = bos + 1
            rows.append({"student_input_ids": ids.tolist(), "student_attention_mask": [1] * n,
                         "teacher_input_ids": ids.tolist(), "teacher_attention_mask": [1] * n})
        n_eval = min(cfg.eval_samples, max(0, len(rows) // 5)) if cfg.save_strategy != "no" else 0
        dataset, eval_dataset = rows[:len(rows) - n_eval], (rows[len(rows) - n_eval:] if n_eval else None)
    else:
        from datasets import load_from_disk
        dataset = load_from_disk(cfg.dataset_path)     # train.py:234-236 (pre-processed columns, data.py:124-141)
        if isinstance(dataset, dict):                  # train.py:243-247: a DatasetDict -> its train split
            dataset = dataset.get("train", dataset)
        if not {"student_input_ids", "teacher_input_ids"} <= set(dataset.column_names):   # train.py:253-256
            raise SystemExit("this path takes PRE-PROCESSED rows (student_input_ids / teacher_input_ids [+ teacher_top_k_v/i]); "
                             "the on-the-fly DistillDataProcessor of train.py:279-328 is outside the hot path")
        eval_dataset = None
        if cfg.save_strategy != "no":                  # train.py:262-269
            split = dataset.train_test_split(test_size=min(cfg.eval_samples, max(1, len(dataset) // 5)), seed=42)
            dataset, eval_dataset = split["train"], split["test"]
    evaluate = eval_dataset is not None
    args = TrainingArguments(
        output_dir=cfg.output_dir, per_device_train_batch_size=cfg.per_device_train_batch_size,
        gradient_accumulation_steps=cfg.gradient_accumulation_steps, num_train_epochs=cfg.epochs,
        learning_rate=cfg.learning_rate, logging_steps=cfg.logging_steps, bf16=True, gradient_checkpointing=True,
        eval_strategy="epoch" if evaluate else "no", save_strategy=cfg.save_strategy,
        load_best_model_at_end=evaluate and cfg.save_strategy == "epoch", save_total_limit=3,
        report_to=[] if cfg.report_to in ("none", "") else cfg.report_to,
        remove_unused_columns=False, label_names=["labels"], max_steps=cfg.max_steps, warmup_steps=cfg.warmup_steps,
        dataloader_num_workers=cfg.dataloader_num_workers,                                  # train.py:348-353
        dataloader_prefetch_factor=cfg.dataloader_prefetch_factor if cfg.dataloader_num_workers > 0 else None,
        dataloader_pin_memory=True, seed=cfg.seed, logging_nan_inf_filter=cfg.logging_nan_inf_filter,
        **({} if cfg.blocking_h2d else {"accelerator_config": {"non_blocking": True}}),
        **({"ddp_backend": cfg.ddp_backend} if cfg.ddp_backend else {}))     # train.py:331-354
    # Under torchrun the trainer wraps the student in ddp.HipDataParallel itself (DistillationTrainer._wrap_model):
    # bucketed RCCL all-reduce of the flat gradient under the backward, no_sync on accumulation micro-batches.
    trainer = DistillationTrainer(model=student, args=args, train_dataset=dataset, eval_dataset=eval_dataset,
                                  data_collator=ProcessedDataCollator(tokenizer, speech_bos=cfg.speech_bos, pad_token_id=pad),
                                  teacher_model=teacher, temperature=cfg.temperature, alpha=cfg.alpha, top_k=cfg.top_k,
                                  is_quantized_teacher=cfg.load_teacher_in_4bit or cfg.load_teacher_in_8bit)
    t0 = time.time()
    if os.environ.get("SD_PROFILE_HOST"):              # (+) where the host spends its time: cProfile, top of cumulative
        import cProfile
        import pstats
        prof = cProfile.Profile()
        prof.runcall(trainer.train)
        pstats.Stats(prof).sort_stats("cumulative").print_stats(45)
    else:
        trainer.train()                                # train.py:420
    if trainer.is_world_process_zero():
        print(f"done in {time.time() - t0:.1f}s; log tail: {trainer.state.log_history[-3:]}")
    if cfg.log_json and (trainer.is_world_process_zero() or "{rank}" in cfg.log_json):
        import json
        red = getattr(student, "_reducer", None)
        with open(cfg.log_json.replace("{rank}", os.environ.get("RANK", "0")), "w") as f:
            json.dump({"log_history": trainer.state.log_history, "best_model_checkpoint": trainer.state.best_model_checkpoint,
                       "world_size": int(os.environ.get("WORLD_SIZE", 1)), "wrapped": type(trainer.model_wrapped).__name__,
                       "optimizer": type(getattr(trainer.optimizer, "optimizer", trainer.optimizer)).__name__,
                       "reducer": None if red is None else red.stats, "global_step": trainer.state.global_step,
                       "param_checksum": float(student.flat.double().sum()),
                       "adapter_checksum": None if student._lora is None else float(student._lora.master.double().abs().sum()),
                       "trainable": sorted({n.split(".")[-2] for n, q in student.named_parameters() if q.requires_grad})}, f)


if __name__ == "__main__":
    main()
