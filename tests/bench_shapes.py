"""GPU micro-benchmark (not a pytest): times every GEMM shape of one BASELINE-config-2 step and the
row kernels, one launch kind at a time, with torch.cuda events.  Output: gpurun_out/shapes.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import _lib, ops  # noqa: E402

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
    return a.elapsed_time(b) / iters * 1e3  # us


def main():
    M, V = 2048, 159488
    shapes = []
    for tag, h, I in (("student", 1024, 3072), ("teacher", 2048, 6144)):
        for name, N, K in (("qkv", 4096, h), ("o", h, 2048), ("gu", 2 * I, h), ("down", h, I), ("lm_head", V, h)):
            shapes.append((f"{tag}.{name}.fwd_NT", M, N, K, False, False, False))
            if tag == "student":
                shapes.append((f"{tag}.{name}.dX_NN", M, K, N, False, True, name == "lm_head"))
                shapes.append((f"{tag}.{name}.dW_TN", N, K, M, True, True, False))
    out = []
    g = torch.Generator(device=dev).manual_seed(0)
    lib = ops.load_lib()
    variants = [(0, 0), (128, 2), (128, 3), (128, 4), (64, 2), (64, 3), (64, 4), (256, 9), (64, 9)] if "--tune" in sys.argv else [(0, 0)]
    for name, m, n, k, ta, tb, sk in shapes:
        a = torch.randn((k, m) if ta else (m, k), device=dev, generator=g).bfloat16()
        b = torch.randn((k, n) if tb else (n, k), device=dev, generator=g).bfloat16()
        c = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        # in the real step every layer has its own weights (3.5 GB teacher + 1.2 GB student per step >> 256 MB
        # Infinity Cache): rotate through enough copies of the weight-side operand that it is always HBM-cold
        wcold = "--cold" in sys.argv
        w_is_b = not (ta and tb)          # NT/NN: B is the weight; TN (dW): both operands are activations
        ncopy = max(2, int(600e6 // (b.numel() * 2)) + 1) if (wcold and w_is_b) else 1
        bs = [b] + [b.clone() for _ in range(ncopy - 1)]
        rot = [0]

        def run(use_sk):
            rot[0] = (rot[0] + 1) % ncopy
            ops.gemm(a, bs[rot[0]], ta, tb, out=c, split_k=use_sk)
        row = {"name": name, "M": m, "N": n, "K": k}
        line = f"{name:26s} M={m:6d} N={n:6d} K={k:6d} "
        for bm, nst in variants:
            for use_sk in ([False, True] if ("--tune" in sys.argv and k >= 2048 and (m // 64) * (n // 128) < 512) else [sk]):
                _lib.gemm_force_variant(bm, nst)
                us = timeit(lambda: run(use_sk), iters=8 if n > 100000 or k > 100000 else 25)
                tf = 2.0 * m * n * k / us / 1e6
                key = ("auto" if bm == 0 else f"{bm}x{nst & 0xff}" + ("chk" if nst & 0x100 else "")) + ("+sk" if use_sk else "")
                row[key] = round(tf, 1)
                row[key + "_us"] = round(us, 1)
                line += f" {key}:{tf:6.0f}"
        _lib.gemm_force_variant(0, 0)
        out.append(row)
        print(line, flush=True)
        del a, b, c, bs
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/shapes.json", "w"), indent=1)


if __name__ == "__main__":
    main()
