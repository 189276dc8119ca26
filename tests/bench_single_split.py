"""GPU micro-benchmark (not a pytest): ONE forward projection at M = 2048 on the current kernel choice against the
persistent 256x128 kernel with the contraction cut into K slices (sd_gemm_grouped_nt with one problem; fp32 slabs
left for the consumer).  --cold: 512 MiB written between launches (operands leave the Infinity Cache)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import ops  # noqa: E402
from bench_pair import timeit  # noqa: E402

dev = torch.device("cuda:0")


def main():
    M = 2048
    flush = torch.zeros(128 * 1024 * 1024, device=dev) if "--cold" in sys.argv else None
    g = torch.Generator(device=dev).manual_seed(0)
    for name, N, K in (("teacher o", 2048, 2048), ("teacher down", 2048, 6144), ("student o", 1024, 2048),
                       ("student down", 1024, 3072), ("teacher qkv (no rope)", 4096, 2048)):
        x = (torch.randn(M, K, device=dev, generator=g)).bfloat16()
        w = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
        r = torch.randn(M, N, device=dev, generator=g).bfloat16()
        fl = 2.0 * M * N * K
        t = timeit(lambda: ops.gemm(x, w, residual=r), flush=flush)
        print(f"{name:22s} current (+residual)      {t:7.1f} us  {fl / t / 1e6:6.0f} TF/s", flush=True)
        for ns in (1, 2, 3, 4, 6, 8):
            if (K // 64) // ns < 4:
                continue
            t = timeit(lambda: ops.gemm_grouped_nt([(x, w, ns)]), flush=flush)
            print(f"{name:22s} persistent, {ns} K slices   {t:7.1f} us  {fl / t / 1e6:6.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
