"""GPU micro-benchmark (not a pytest): the forward GEMMs that take the persistent 256x256 kernel, cold weights
(a 300 MB flush between launches is NOT done here: these shapes re-stream more than the L2 anyway).
SD_GEMM_NO_P256=1 gives the 256x128 persistent kernel for comparison."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import ops  # noqa: E402

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
    tag = "256x128" if os.environ.get("SD_GEMM_NO_P256") else "256x256"
    for name, M, N, K in (("student gate|up", 2048, 6144, 1024), ("teacher gate|up (plain)", 2048, 12288, 2048),
                          ("student lm_head", 1536, 159488, 1024), ("teacher lm_head", 1536, 159488, 2048)):
        a = torch.randn(M, K, device=dev).bfloat16()
        b = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemm(a, b, out=out), iters=10 if N > 100000 else 30)
        print(f"[{tag}] {name:26s} M={M} N={N} K={K}: {t:8.1f} us  {2.0 * M * N * K / t / 1e6:7.0f} TF/s", flush=True)
    for name, M, I, K in (("student gate|up + SwiGLU", 2048, 3072, 1024), ("teacher gate|up + SwiGLU", 2048, 6144, 2048)):
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(2 * I, K, device=dev) * 0.02).bfloat16()
        t = timeit(lambda: ops.gemm_swiglu(x, w, save_gu=False), iters=30)
        print(f"[{tag}] {name:26s} M={M} I={I} K={K}: {t:8.1f} us  {4.0 * M * I * K / t / 1e6:7.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
