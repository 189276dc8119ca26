"""GPU micro-benchmark (not a pytest): attention forward, software-pipelined kernel (default) against the classic one
(sd_attn_force_variant(1)) at B*T = 2048 tokens and larger, outputs compared bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import ops  # noqa: E402
from speech_distill_amd._lib import load_lib  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=100, warm=10):
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
    lib = load_lib()
    Hq, Hkv = 16, 8
    for B, T in ((4, 512), (8, 256), (2, 1024), (1, 2048), (16, 512), (4, 2048)):
        M = B * T
        qkv = torch.randn(M, (Hq + 2 * Hkv) * 128, device=dev).bfloat16()
        q, k, v = qkv[:, :Hq * 128], qkv[:, Hq * 128:(Hq + Hkv) * 128], qkv[:, (Hq + Hkv) * 128:]
        res = {}
        for name, variant in (("pipe", 2), ("classic", 1), ("pipe", 2), ("classic", 1)):
            _lib.debug_set("attn.variant", variant)
            o, lse = ops.attn_fwd(q, k, v, B, T, Hq, Hkv)
            t = timeit(lambda: ops.attn_fwd(q, k, v, B, T, Hq, Hkv))
            res.setdefault(name, []).append((t, o, lse))
        _lib.debug_set("attn.variant", 0)
        same = torch.equal(res["pipe"][0][1], res["classic"][0][1]) and torch.equal(res["pipe"][0][2], res["classic"][0][2])
        fl = 2.0 * B * Hq * T * T * 128
        tp = min(x[0] for x in res["pipe"])
        tc = min(x[0] for x in res["classic"])
        print(f"B={B:3d} T={T:5d}: pipe {tp:7.1f} us ({fl / tp / 1e6:6.0f} TF/s)  classic {tc:7.1f} us ({fl / tc / 1e6:6.0f} TF/s)  "
              f"identical={same}", flush=True)


if __name__ == "__main__":
    main()
