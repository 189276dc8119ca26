"""GPU micro-benchmark (not a pytest): the frozen teacher's forward + top-K per token at B = 4 (one micro-batch, as the
distillation step runs it) against B = 8 / 16 (two / four micro-batches of a gradient-accumulation window at once), alone
on the chip, head on 75 % of the rows as in the step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_distill_amd as sda  # noqa: E402
from speech_distill_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
teacher = sda.HipQwen3ForCausalLM(sda.Qwen3Dims.teacher_17b(), device=dev, init_std=0)
teacher.flat.normal_(0.0, 0.02, generator=torch.Generator(device=dev).manual_seed(1))
for p in teacher._params.values():
    if p.dim() == 1:
        p.data.fill_(1.0)
teacher.eval().requires_grad_(False)
T = 512
for B in (4, 8, 16, 4, 16):
    ids = torch.randint(0, 159488, (B, T), device=dev)
    labels = ids.clone()
    labels[:, :128] = -100
    rows, _ = ops.loss_rows(labels)

    def f():
        with torch.no_grad():
            return ops.logsoftmax_topk(teacher(input_ids=ids, logit_rows=rows).logits, 128, 159488)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"B={B:3d}: {dt * 1e3:7.2f} ms  = {dt / (B * T) * 1e6:6.3f} us per token  ({dt * 1e3 * 4 / B:6.2f} ms per 2048 tokens)", flush=True)
