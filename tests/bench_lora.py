"""Measurement: the LoRA student (train.py:180-202, speech_distill_amd/lora.py) at the BASELINE config-2 shapes
(0.6B student, r = 32 on the seven projections of 28 layers, 1.7B teacher, B = 4, T = 512).

  * duration and HBM rate of the three table-driven launches of csrc/sd_lora.hip over all 196 targets
    (algorithmic bytes: merge 4 B / projection weight, dB and dA 2 B each) and of the optimizer step;
  * a training loop with the reference's gradient accumulation (train.py:524: 4 micro-batches per optimizer step),
    LoRA student vs fully trained student, same micro-step kernels;
  * optionally (--pissa) the time of the PiSSA initialisation (196 SVDs on the GPU).

    python tests/bench_lora.py [--pissa] [--steps 6] > gpurun_out/lora_bench.json
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import VOCAB, synthetic_batch  # noqa: E402


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pissa", action="store_true")
    ap.add_argument("--steps", type=int, default=6, help="optimizer steps timed (each = --accum micro-batches)")
    ap.add_argument("--accum", type=int, default=4)
    ap.add_argument("--rank", type=int, default=32)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    import speech_distill_amd as sda
    from speech_distill_amd import lora as L, ops
    from speech_distill_amd._lib import check, load_lib
    from speech_distill_amd.optim import FlatAdamW
    lib = load_lib()

    def fresh(dims, seed):
        m = sda.HipQwen3ForCausalLM(dims, device=dev, init_std=0)
        m.flat.normal_(0.0, 0.02, generator=torch.Generator(device=dev).manual_seed(seed))
        for p in m._params.values():
            if p.dim() == 1:
                p.data.fill_(1.0)
        return m
    teacher = fresh(sda.Qwen3Dims.teacher_17b(), 1)
    teacher.eval().requires_grad_(False)
    out = {"config": {"student": "qwen3-0.6B shape", "r": args.rank, "targets": 7 * 28, "B": 4, "T": 512, "accum": args.accum}}

    t0 = time.perf_counter()
    student = L.get_lora_model(fresh(sda.Qwen3Dims.student_06b(), 0),
                               L.LoraConfig(r=args.rank, init_lora_weights="pissa" if args.pissa else "gaussian"))
    torch.cuda.synchronize()
    out["attach_s"] = {"init": "pissa" if args.pissa else "gaussian", "seconds": round(time.perf_counter() - t0, 2)}
    st = student._lora
    train, total = st.trainable_parameters()
    out["trainable"] = {"params": train, "all": total, "adapter": int(sum(p.numel() for p in st.params.values()))}
    with torch.no_grad():   # a live adapter
        st.master[st.n_a:].normal_(0.0, 0.01)
    st.refresh_shadows()
    stream = torch.cuda.current_stream().cuda_stream
    nw = st.base.numel()
    ms = timed(lambda: check(lib.sd_lora_merge(st._plan_dev.data_ptr(), st._plan_host, stream), "merge"))
    out["merge"] = {"ms": round(ms, 4), "bytes": 4 * nw, "GBps": round(4 * nw / ms / 1e6, 1), "frac_of_8TBps": round(4 * nw / ms / 1e6 / 8000, 3)}
    ops.prof_begin()
    for _ in range(5):
        check(lib.sd_lora_project(st._plan_dev.data_ptr(), st._plan_host, stream), "project")
    ops.prof_end()
    for sym, v in ops.prof_symbols().items():
        if sym.startswith("lora_d"):
            each = v[0] / v[2]
            out[sym.split("<")[0]] = {"ms": round(each, 4), "bytes": 2 * nw, "GBps": round(2 * nw / each / 1e6, 1),
                                      "frac_of_8TBps": round(2 * nw / each / 1e6 / 8000, 3)}
    ms = timed(lambda: check(lib.sd_lora_project(st._plan_dev.data_ptr(), st._plan_host, stream), "project"))
    out["project_both"] = {"ms": round(ms, 4), "bytes": 4 * nw, "GBps": round(4 * nw / ms / 1e6, 1)}

    loss_fn = sda.DistillationLoss(temperature=2.0, alpha=0.5, inplace_grad=True)
    batch = synthetic_batch(4, 512, 0, dev)
    side = ops.concurrent_stream(dev, "teacher")

    def micro(model):
        rows, row_labels = ops.loss_rows(batch["labels"])
        with torch.no_grad():
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                tl = teacher(input_ids=batch["teacher_input_ids"], attention_mask=batch["teacher_attention_mask"],
                             logit_rows=rows, concurrent=True).logits
                tv, ti = ops.logsoftmax_topk(tl, 128, VOCAB)
        logits = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], logit_rows=rows, concurrent=True).logits
        torch.cuda.current_stream().wait_stream(side)
        total_loss = loss_fn.forward_rows(logits, row_labels, teacher_top_k_v=tv, teacher_top_k_i=ti)[0]
        (total_loss / args.accum).backward()
        return total_loss

    def loop(model, opt, steps):
        last = None
        for _ in range(steps):
            for _ in range(args.accum):
                last = micro(model)
            opt.grad_norm(1.0)
            opt.step()
            model.zero_grad()
        return last

    def run(model, label):
        opt = FlatAdamW(model, lr=5e-5)
        loop(model, opt, 2)
        torch.cuda.synchronize()
        t = time.perf_counter()
        last = loop(model, opt, args.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        tok = args.steps * args.accum * 4 * 512
        # the optimizer-step tail alone (projection / norm / update / merge), back to back
        def tail():
            if model._lora is not None:
                model._lora.grads_stale = True
            opt.grad_norm(1.0)
            opt.step()
            if model._lora is not None:
                model._lora.ensure_merged()
        out[label] = {"tokens_per_s": round(tok / dt, 1), "ms_per_micro_step": round(dt / (args.steps * args.accum) * 1e3, 3),
                      "optimizer_tail_ms": round(timed(tail, reps=10, warm=2), 3), "loss": float(last)}

    run(student, "lora_loop")
    del student, st
    torch.cuda.empty_cache()
    run(fresh(sda.Qwen3Dims.student_06b(), 0), "full_loop")
    out["lora_over_full"] = round(out["lora_loop"]["tokens_per_s"] / out["full_loop"]["tokens_per_s"], 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
