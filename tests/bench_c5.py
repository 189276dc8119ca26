"""BASELINE config 5 timing (not a bench.py line): teacher-only forward, batch 64 x seq_len 512, + top-K.
Also config 4-shaped step timing (T=2048, batch 4).  Prints one JSON line each."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_distill_amd as sda
from speech_distill_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, n=5, w=2):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def init(m, seed):
    m.flat.normal_(0.0, 0.02, generator=torch.Generator(device=dev).manual_seed(seed))
    for p in m._params.values():
        if p.dim() == 1:
            p.data.fill_(1.0)


teacher = sda.HipQwen3ForCausalLM(sda.Qwen3Dims.teacher_17b(), device=dev, init_std=0)
init(teacher, 1)
teacher.eval().requires_grad_(False)
teacher.fold_norm_gains = "--no-fold" not in sys.argv  # A/B: RMSNorm gains folded into the weights (round 4) or launched
ids = torch.randint(0, 159488, (64, 512), device=dev)


def c5():
    with torch.no_grad():
        ops.logsoftmax_topk(teacher(input_ids=ids).logits, 100)


dt = timeit(c5)
f = sda.Qwen3Dims.teacher_17b().flops_per_token(512)
print(json.dumps({"config": "C5 teacher-only B=64 T=512 + top-100", "tokens_per_s": 64 * 512 / dt, "ms": dt * 1e3,
                  "mfma_frac": 64 * 512 / dt * f / 2.5e15}), flush=True)
del ids
student = sda.HipQwen3ForCausalLM(sda.Qwen3Dims.student_06b(), device=dev, init_std=0)
init(student, 0)
B, T = 4, 2048
ids = torch.randint(0, 159488, (B, T), device=dev)
labels = ids.clone()
labels[:, : T // 4] = -100
loss_fn = sda.DistillationLoss(2.0, 0.5, inplace_grad=True)


side = ops.concurrent_stream(dev, "teacher")


def c4():  # the sequence of DistillationTrainer.compute_loss on a training step: loss rows only, teacher on a side stream
    student.zero_grad()
    rows, row_labels = ops.loss_rows(labels)
    side.wait_stream(torch.cuda.current_stream())
    with torch.no_grad(), torch.cuda.stream(side):
        tv, ti = ops.logsoftmax_topk(teacher(input_ids=ids, logit_rows=rows).logits, 128)
    logits = student(input_ids=ids, logit_rows=rows).logits
    torch.cuda.current_stream().wait_stream(side)
    loss_fn.forward_rows(logits, row_labels, teacher_top_k_v=tv, teacher_top_k_i=ti)[0].backward()


dt = timeit(c4, n=3, w=1)
f = 3 * sda.Qwen3Dims.student_06b().flops_per_token(T) + sda.Qwen3Dims.teacher_17b().flops_per_token(T)
print(json.dumps({"config": "C4-shaped step B=4 T=2048 (1 GPU), head on the loss rows, teacher beside the student", "tokens_per_s": B * T / dt, "ms": dt * 1e3,
                  "mfma_frac": B * T / dt * f / 2.5e15}), flush=True)

# the same step with gradient checkpointing taken literally (layer-granular recompute, policy "always")
student.gradient_checkpointing_enable(gradient_checkpointing_kwargs={"recompute": "always"})
torch.cuda.synchronize()
torch.cuda.reset_peak_memory_stats()
dt2 = timeit(c4, n=3, w=1)
print(json.dumps({"config": "C4-shaped step, gradient checkpointing = layer recompute", "tokens_per_s": B * T / dt2, "ms": dt2 * 1e3,
                  "vs_saved_activations": dt2 / dt, "peak_gib": torch.cuda.max_memory_allocated() / 2**30}), flush=True)
