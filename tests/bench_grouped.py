"""GPU micro-benchmark (not a pytest): the four dW GEMMs of a student layer, separately vs as one grouped launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=30, warm=3):
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
    K = 2048
    shapes = [(4096, 1024), (1024, 2048), (6144, 1024), (1024, 3072)]
    pairs = [(torch.randn(K, m, device=dev).bfloat16(), torch.randn(K, n, device=dev).bfloat16()) for m, n in shapes]
    outs = [torch.empty(m, n, device=dev, dtype=torch.bfloat16) for m, n in shapes]
    fl = sum(2.0 * K * m * n for m, n in shapes)

    def separate():
        for (a, b), c in zip(pairs, outs):
            ops.gemm(a, b, True, True, out=c)
    t_sep = timeit(separate)
    t_grp = timeit(lambda: ops.gemm_grouped_tn(pairs))
    print(f"separate {t_sep:7.1f} us ({fl / t_sep / 1e6:5.0f} TF/s)   grouped {t_grp:7.1f} us ({fl / t_grp / 1e6:5.0f} TF/s)")


if __name__ == "__main__":
    main()
