"""GPU micro-benchmark + check (not a pytest): gemm_p1_kernel (one compute wave per SIMD) against gemm_pstag_kernel on the
persistent-eligible forward shapes: bit-identical outputs, interleaved timing in one process."""
import os
import sys

import torch

os.environ.setdefault("SD_GEMM_NO_P256", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
lib = ops.load_lib()
M = 2048
g = torch.Generator(device=dev).manual_seed(0)
flush = torch.zeros(128 * 1024 * 1024, device=dev) if "--cold" in sys.argv else None


def run(fn, iters=20):
    ts = []
    for _ in range(iters):
        if flush is not None:
            flush.add_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


for name, K, N, sw in (("teacher gate|up swiglu", 2048, 12288, True), ("teacher gate|up plain", 2048, 12288, False),
                       ("student gate|up plain", 1024, 6144, False), ("lm_head teacher 1536 rows", 2048, 159488, False),
                       ("small M=300", 192, 12288, False)):
    Mx = 1536 if "lm_head" in name else (300 if "small" in name else M)
    x = torch.randn(Mx, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
    fl = 2.0 * Mx * N * K

    def f():
        return ops.gemm_swiglu(x, w, save_gu=False)[0] if sw else ops.gemm(x, w)
    lib.sd_gemm_force_variant(0, 0)
    ref = f()
    lib.sd_gemm_force_variant(0, 0x200)
    got = f()
    same = torch.equal(ref, got)
    res = {}
    for rnd in range(3):
        for tag, v in (("pstag", 0), ("p1", 0x200)):
            lib.sd_gemm_force_variant(0, v)
            f()
            res.setdefault(tag, []).append(run(f))
    lib.sd_gemm_force_variant(0, 0)
    med = {k: sorted(x[0] for x in v)[1] for k, v in res.items()}
    mn = {k: min(x[1] for x in v) for k, v in res.items()}
    print(f"{name:28s} identical={same}  pstag {med['pstag']:7.1f} us (min {mn['pstag']:6.1f}, {fl / med['pstag'] / 1e6:5.0f} TF/s)   "
          f"p1 {med['p1']:7.1f} us (min {mn['p1']:6.1f}, {fl / med['p1'] / 1e6:5.0f} TF/s)", flush=True)
