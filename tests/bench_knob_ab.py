"""Measurement: A/B of one sd_debug knob on the config-2 micro-step INSIDE one process (same box, same clocks, the two
settings alternating in blocks), because box-to-box and run-to-run spread (0.3-1 ms) hides a 1 % effect.

    python tests/bench_knob_ab.py gemm.fwd_bump 0 3 [--rounds 8] [--block 10]
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import VOCAB, synthetic_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("key")
    ap.add_argument("values", nargs="+", type=int)
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--block", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--seq-len", type=int, default=512)
    ap.add_argument("--student-first", action="store_true", help="enqueue the student's forward before the teacher's")
    ap.add_argument("--no-fold", action="store_true", help="the teacher runs its RMSNorm launches (no folded gains)")
    ap.add_argument("--cached-rows", action="store_true", help="select the loss rows once (no host read per step)")
    ap.add_argument("--main-priority", type=int, default=0, help="run the step on a stream of this priority (-1 = high)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    import speech_distill_amd as sda
    from speech_distill_amd import _lib, ops

    def fresh(dims, seed):
        m = sda.HipQwen3ForCausalLM(dims, device=dev, init_std=0)
        m.flat.normal_(0.0, 0.02, generator=torch.Generator(device=dev).manual_seed(seed))
        for p in m._params.values():
            if p.dim() == 1:
                p.data.fill_(1.0)
        return m
    teacher, student = fresh(sda.Qwen3Dims.teacher_17b(), 1), fresh(sda.Qwen3Dims.student_06b(), 0)
    teacher.eval().requires_grad_(False)
    teacher.fold_norm_gains = not args.no_fold
    loss_fn = sda.DistillationLoss(temperature=2.0, alpha=0.5, inplace_grad=True)
    batch = synthetic_batch(args.batch, args.seq_len, 0, dev)
    side = ops.concurrent_stream(dev, "teacher")

    cache = []

    def micro():
        student.zero_grad()
        if args.cached_rows and cache:
            rows, row_labels = cache[0]
        else:
            rows, row_labels = ops.loss_rows(batch["labels"])
            cache[:] = [(rows, row_labels)]
        def run_teacher():
            with torch.no_grad():
                with torch.cuda.stream(side):
                    tl = teacher(input_ids=batch["teacher_input_ids"], attention_mask=batch["teacher_attention_mask"],
                                 logit_rows=rows, concurrent=True).logits
                    return ops.logsoftmax_topk(tl, 128, VOCAB)
        side.wait_stream(torch.cuda.current_stream())
        if not args.student_first:
            tv, ti = run_teacher()
        logits = student(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], logit_rows=rows, concurrent=True).logits
        if args.student_first:
            tv, ti = run_teacher()
        torch.cuda.current_stream().wait_stream(side)
        loss_fn.forward_rows(logits, row_labels, teacher_top_k_v=tv, teacher_top_k_i=ti)[0].backward()

    if args.main_priority:
        torch.cuda.set_stream(torch.cuda.Stream(priority=args.main_priority))
    for _ in range(5):
        micro()
    res = {v: [] for v in args.values}
    for r in range(args.rounds):
        order = args.values if r % 2 == 0 else args.values[::-1]
        for v in order:
            _lib.debug_set(args.key, v)
            micro()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(args.block):
                micro()
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t) / args.block * 1e3)
    _lib.debug_set(args.key, 0)
    out = {"key": args.key, "ms_per_step": {str(v): {"median": round(statistics.median(x), 4), "min": round(min(x), 4),
                                                     "all": [round(y, 3) for y in x]} for v, x in res.items()}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
