"""GPU micro-benchmark (not a pytest): the HBM-bound kernels of the step at BASELINE config 2 sizes -- teacher
log-softmax + top-128, loss forward / backward on the 1536 loss rows, RMSNorm -- as algorithmic GB/s.
SD_TOPK_NT=256|512|1024 selects the top-K workgroup size (A/B)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import ops  # noqa: E402
from speech_distill_amd.distillation_loss import DistillationLoss  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    R, V, K = 1536, 159488, 128
    logits = (torch.randn(R, V, device=dev) * 2).bfloat16()
    t = timeit(lambda: ops.logsoftmax_topk(logits, K, V))
    print(f"topk  R={R} V={V} K={K} NT={os.environ.get('SD_TOPK_NT', 'default')}: {t:7.1f} us  {R * V * 2 / t / 1e3:7.0f} GB/s", flush=True)
    tv, ti = ops.logsoftmax_topk(logits, K, V)
    labels = torch.randint(0, V, (R,), device=dev)
    s = (torch.randn(R, V, device=dev) * 2).bfloat16().requires_grad_(True)
    fn = DistillationLoss(2.0, 0.5)

    def fwd():
        return fn.forward_rows(s, labels, teacher_top_k_v=tv, teacher_top_k_i=ti)[0]
    tf = timeit(fwd)
    print(f"loss fwd rows: {tf:7.1f} us  {R * V * 2 / tf / 1e3:7.0f} GB/s", flush=True)

    def fb():
        s.grad = None
        fwd().backward()
    tfb = timeit(fb)
    print(f"loss fwd+bwd rows: {tfb:7.1f} us  (bwd ~{tfb - tf:6.1f} us, {R * V * 4 / max(tfb - tf, 1) / 1e3:7.0f} GB/s)", flush=True)
    for M, H in ((2048, 1024), (2048, 2048)):
        x = torch.randn(M, H, device=dev).bfloat16()
        w = torch.ones(H, device=dev).bfloat16()
        tn = timeit(lambda: ops.rmsnorm_fwd(x, w), iters=50)
        print(f"rmsnorm fwd M={M} H={H}: {tn:6.1f} us  {M * H * 4 / tn / 1e3:7.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
