"""GPU measurement (not a pytest): every GEMM call of the distillation step at the BASELINE shapes -- config 2 (2 048 tokens,
heads on 1 536 rows), config 4 (8 192 tokens, heads on 6 144 rows), config 5 (teacher only, 32 768 tokens) -- under every
kernel variant the dispatcher can pick (forced through include/sd_hip_debug.h), weight operand HBM-cold.  Output:
gpurun_out/gemm_tune.json = per call the time of the heuristic's choice ("auto") and of each forced variant.
scripts/make_gemm_table.py turns it into speech_distill_amd/csrc/sd_gemm_table.inc."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
V = 159488


def timeit(fn, iters, warm=2):
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


def calls(cfg):
    """(name, form, epi, M, N, K) -- N, K in GEMM terms (C [M,N], contraction K)"""
    M, R = {"c2": (2048, 1536), "c4": (8192, 6144), "c5": (32768, 32768)}[cfg]
    out = []
    models = (("teacher", 2048, 6144),) if cfg == "c5" else (("student", 1024, 3072), ("teacher", 2048, 6144))
    for tag, h, inter in models:
        out += [(f"{cfg}.{tag}.qkv", "NT", 4, M, 4096, h), (f"{cfg}.{tag}.o", "NT", 1, M, h, 2048),
                (f"{cfg}.{tag}.gu", "NT", 3 if tag == "teacher" else 0, M, 2 * inter, h),
                (f"{cfg}.{tag}.down", "NT", 1, M, h, inter)]
        if cfg != "c5":
            out.append((f"{cfg}.{tag}.lm_head", "NT", 0, R, V, h))
        else:
            out.append((f"{cfg}.{tag}.lm_head", "NT", 0, 8192, V, h))  # scripts/extract_teacher_logits.py: whole batch; one slice
        if tag == "student":
            out += [(f"{cfg}.{tag}.lm_head.dW", "TN", 0, V, h, R), (f"{cfg}.{tag}.lm_head.dX", "NN", 2, R, h, V),
                    (f"{cfg}.{tag}.down.dX", "NN", 5, M, inter, h), (f"{cfg}.{tag}.gu.dX", "NN", 2, M, h, 2 * inter),
                    (f"{cfg}.{tag}.o.dX", "NN", 6, M, 2048, h), (f"{cfg}.{tag}.qkv.dX", "NN", 2, M, h, 4096)]
    return out


# "warm" is measured first and discarded (first touches, allocator growth); "auto" (the dispatcher's own choice) runs LAST
VARIANTS = [("warm", 0, 0, {}), ("64x3", 64, 3, {}), ("64x4", 64, 4, {}), ("128x2", 128, 2, {}), ("128x3", 128, 3, {}),
            ("128x4", 128, 4, {}), ("256x3", 256, 3, {}), ("256x9", 256, 9, {}), ("256x9.nopersist", 256, 9, {"gemm.no_persist": 1}),
            ("256x9.nop256", 256, 9, {"gemm.no_p256": 1}), ("256x9.p256", 256, 9, {"gemm.p256_min_tiles": 1}), ("auto", 0, 0, {})]


def main():
    cfgs = [a for a in sys.argv[1:] if a in ("c2", "c4", "c5")] or ["c2", "c4", "c5"]
    g = torch.Generator(device=dev).manual_seed(0)
    rows = []
    for cfg in cfgs:
        for name, form, epi, m, n, k in calls(cfg):
            ta, tb = form == "TN", form in ("NN", "TN")
            a = torch.randn((k, m) if ta else (m, k), device=dev, generator=g).bfloat16()
            b = torch.randn((k, n) if tb else (n, k), device=dev, generator=g).bfloat16()
            ncopy = 1 if form == "TN" else max(2, int(600e6 // (b.numel() * 2)) + 1)
            bs = [b] + [b.clone() for _ in range(ncopy - 1)]
            rot = [0]
            T = 512 if cfg != "c4" else 2048
            extra = {}
            if epi == 1:
                extra["r"] = torch.randn(m, n, device=dev, generator=g).bfloat16()
            if epi == 4:
                extra["qg"] = torch.ones(128, device=dev, dtype=torch.bfloat16)
                extra["cs"] = ops.rope_tables(T, dev)
            if epi == 5:
                extra["gu"] = torch.randn(m, 2 * n, device=dev, generator=g).bfloat16()
            if epi == 6:
                extra["o"] = torch.randn(m, n, device=dev, generator=g).bfloat16()

            def run():
                rot[0] = (rot[0] + 1) % ncopy
                w = bs[rot[0]]
                if epi == 1:
                    ops.gemm(a, w, residual=extra["r"])
                elif epi == 3:
                    ops.gemm_swiglu(a, w, save_gu=False)
                elif epi == 4:
                    ops.gemm_qkv_rope(a, w, extra["qg"], extra["qg"], extra["cs"][0], extra["cs"][1], T, 16, 8)
                elif epi == 5:
                    ops.gemm_swiglu_bwd(a, w, extra["gu"])
                elif epi == 6:
                    ops.gemm_odx_delta(a, w, extra["o"], T, 16)
                elif epi == 2:
                    ops.gemm(a, w, ta, tb, split_k=True)
                else:
                    ops.gemm(a, w, ta, tb)
            row = {"name": name, "cfg": cfg, "form": form, "epi": epi, "M": m, "N": n, "K": k, "gflop": 2.0 * m * n * k / 1e9, "us": {}}
            iters = 5 if (max(m, n, k) > 100000 or m >= 32768) else 20
            line = f"{name:26s} {form} epi{epi} M={m:6d} N={n:6d} K={k:6d}"
            for vname, bm, nst, dbg in VARIANTS:
                if bm == 256 and form != "NT" and "p256" in vname:
                    continue
                try:
                    for kk, vv in dbg.items():
                        _lib.debug_set(kk, vv)
                    _lib.gemm_force_variant(bm, nst)
                    us = timeit(run, iters)
                    row["us"][vname] = round(us, 2)
                    line += f"  {vname}:{us:.1f}"
                except Exception as e:  # a variant that cannot run this epilogue
                    row["us"][vname] = None
                finally:
                    _lib.debug_set("reset", 0)
            row["us"].pop("warm", None)
            ok = {k_: v for k_, v in row["us"].items() if v}
            best = min(ok, key=ok.get)
            row["best"], row["gain_over_auto"] = best, round(ok["auto"] / ok[best], 3)
            line += f"   BEST {best} x{row['gain_over_auto']:.3f}"
            print(line, flush=True)
            rows.append(row)
            del a, b, bs, extra
            torch.cuda.empty_cache()
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump({"device": torch.cuda.get_device_name(0), "rows": rows}, open("gpurun_out/gemm_tune.json", "w"), indent=1)


if __name__ == "__main__":
    main()
