"""GPU micro-benchmark (not a pytest): forward (NT) GEMM time per 256x128 tile as a function of K and of the number of
tiles per CU, weights HBM-cold -- separates the per-K-step cost from the per-tile cost of the staggered kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=2):
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
    M = 2048
    for N in (4096, 16384, 65536):
        for K in (256, 512, 1024, 2048, 4096):
            a = torch.randn(M, K, device=dev).bfloat16()
            ncopy = max(2, int(600e6 // (N * K * 2)) + 1)
            bs = [torch.randn(N, K, device=dev).bfloat16() for _ in range(ncopy)]
            c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            rot = [0]

            def run():
                rot[0] = (rot[0] + 1) % ncopy
                ops.gemm(a, bs[rot[0]], False, False, out=c)
            us = timeit(run)
            tiles_per_cu = (M // 256) * (N // 128) / 256
            print(f"N={N:6d} K={K:5d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:6.0f} TF/s  tiles/CU {tiles_per_cu:5.1f}  "
                  f"us/tile {us / tiles_per_cu:6.2f}  us/K-step {us / tiles_per_cu / (K / 64):5.2f}", flush=True)
            del bs, a, c


if __name__ == "__main__":
    main()
