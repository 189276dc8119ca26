"""GPU micro-benchmark (not a pytest): the forward projections of one decoder layer of the student AND the teacher at
M = 2048 -- separately (as two launches, back to back on one stream: the sum of what the two streams of the step run)
against ONE grouped persistent launch (sd_gemm_grouped_nt), for several K-slice plans of the N = hidden projections."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=30, warm=5, flush=None):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(iters):
        if flush is not None:
            flush.add_(1.0)  # 512 MiB written: weights and activations leave the Infinity Cache, as inside the step
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    return tot / iters * 1e3


def main():
    M = 2048
    flush = torch.zeros(128 * 1024 * 1024, device=dev) if "--cold" in sys.argv else None
    g = torch.Generator(device=dev).manual_seed(0)

    def rnd(*shape, s=1.0):
        return (torch.randn(*shape, device=dev, generator=g) * s).bfloat16()
    # (name, K_t, N_t, K_s, N_s, swiglu)
    ops_ = (("qkv", 2048, 4096, 1024, 4096, False), ("o", 2048, 2048, 2048, 1024, False),
            ("gate|up", 2048, 12288, 1024, 6144, True), ("down", 6144, 2048, 3072, 1024, False))
    for name, Kt, Nt, Ks, Ns, sw in ops_:
        xt, wt, xs, ws = rnd(M, Kt), rnd(Nt, Kt, s=0.02), rnd(M, Ks), rnd(Ns, Ks, s=0.02)
        flops = 2.0 * M * (Kt * Nt + Ks * Ns)

        def separate():
            if sw:
                ops.gemm_swiglu(xt, wt, save_gu=False)
                ops.swiglu_fwd(ops.gemm(xs, ws))
            else:
                ops.gemm(xt, wt)
                ops.gemm(xs, ws)
        t = timeit(separate, flush=flush)
        print(f"{name:8s} separate launches             {t:7.1f} us  {flops / t / 1e6:6.0f} TF/s", flush=True)
        plans = [(1, 1)] if sw or name == "qkv" else [(1, 1), (2, 1), (2, 2), (3, 2), (4, 2), (3, 1), (4, 4)]
        for nt, ns in plans:
            t = timeit(lambda: ops.gemm_grouped_nt([(xt, wt, nt), (xs, ws, ns)], swiglu=sw), flush=flush)
            print(f"{name:8s} grouped  K slices T{nt} S{ns}        {t:7.1f} us  {flops / t / 1e6:6.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
