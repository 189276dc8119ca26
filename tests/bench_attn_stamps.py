"""GPU diagnostic (not a pytest): where the heaviest workgroup of attn_fwd_pipe_kernel / attn_fwd_kernel spends its cycles.  Needs the stamp build,
made ON the GPU box (the library lives in /tmp):
    make -C speech_distill_amd/csrc stamps && SD_HIP_LIB=/tmp/sd_stamps/libsd_hip.so python tests/bench_attn_stamps.py
Prints s_memtime deltas (100 MHz ticks -> ns) of waves 0 (heavy tile, keys 0-31), 2 (heavy, keys 32-63) and 4 (light)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import ops  # noqa: E402
from speech_distill_amd._lib import load_lib  # noqa: E402

dev = torch.device("cuda:0")


def main():
    lib = load_lib()
    lib.sd_debug_attn_stamp_buffer.argtypes = [C.c_void_p]
    lib.sd_debug_attn_stamp_buffer.restype = None
    buf = torch.zeros(8 * 256, dtype=torch.int64, device=dev)
    Hq, Hkv = 16, 8
    for variant, B, T in ((2, 4, 512), (1, 4, 512), (2, 1, 2048), (1, 1, 2048)):
        _lib.debug_set("attn.variant", variant)
        M = B * T
        qkv = torch.randn(M, (Hq + 2 * Hkv) * 128, device=dev).bfloat16()
        q, k, v = qkv[:, :Hq * 128], qkv[:, Hq * 128:(Hq + Hkv) * 128], qkv[:, (Hq + Hkv) * 128:]
        for _ in range(3):
            ops.attn_fwd(q, k, v, B, T, Hq, Hkv)
        torch.cuda.synchronize()
        lib.sd_debug_attn_stamp_buffer(buf.data_ptr())
        buf.zero_()
        ops.attn_fwd(q, k, v, B, T, Hq, Hkv)
        torch.cuda.synchronize()
        lib.sd_debug_attn_stamp_buffer(None)
        st = buf.cpu().view(8, 256).tolist()
        nkv = T // 64
        t0 = min(st[w][0] for w in range(8))
        print(f"== {'classic' if variant == 1 else 'pipelined'} kernel, B={B} T={T}: nkv={nkv}; shader cycles since the first wave's entry")
        for w in (0, 2, 4):
            r = [x - t0 for x in st[w]]
            print(f" wave {w}: entry {r[0]}  prologue issued {r[1]}")
            shown = list(range(min(nkv, 4))) + ([nkv - 2, nkv - 1] if nkv > 4 else [])
            for i in shown:
                a, b_, c_, d = r[2 + 4 * i: 6 + 4 * i]
                prev = r[1 + 4 * i] if i else r[1]
                print(f"   tile {i:2d}: wait {a - prev:6d}  barrier {b_ - a:6d}  issue {c_ - b_:5d}  compute {d - c_:6d}   (t={d})")
            e = 2 + 4 * nkv
            print(f"   loop left {r[e]}  merged {r[e + 1]}  stores issued {r[e + 2] if st[w][e + 2] else '-'}")
            if st[w][200] and nkv > 3:
                names = ["S^T MFMAs issued", "lane maxima (S^T done)", "row maxima (shuffle)", "exponentials",
                         "rescale", "V^T fragments waited", "P.V MFMAs issued"]
                fine = st[w][200:208]
                print("   inside tile 3 (cycles): " + ", ".join(f"{n} {fine[k + 1] - fine[k]}" for k, n in enumerate(names)))


if __name__ == "__main__":
    main()
