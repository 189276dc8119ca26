"""GPU micro-benchmark (not a pytest): SPATIAL sharing of the chip between the teacher's and the student's forward GEMM
chains.  Today both streams launch kernels sized for all 256 CUs (small tiles for the student's small GEMMs) and the two
streams take turns; here each stream launches persistent 256x128-tile kernels sized for its SHARE of the CUs (grid = share),
so that the two chains run side by side.  28 layers x (qkv, o, gate|up + SwiGLU, down) per model, no attention / norms."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
lib = ops.load_lib()
M, L = 2048, 28
g = torch.Generator(device=dev).manual_seed(0)


def model(h, I):
    per = 4096 * h + h * 2048 + 2 * I * h + h * I
    flat = (torch.randn(L * per, device=dev, generator=g) * 0.02).bfloat16()
    ws = []
    for l in range(L):
        o = l * per
        wqkv = flat[o:o + 4096 * h].view(4096, h); o += 4096 * h
        wo = flat[o:o + h * 2048].view(h, 2048); o += h * 2048
        wgu = flat[o:o + 2 * I * h].view(2 * I, h); o += 2 * I * h
        wd = flat[o:o + h * I].view(h, I)
        ws.append((wqkv, wo, wgu, wd))
    x = torch.randn(M, h, device=dev, generator=g).bfloat16()
    ao = torch.randn(M, 2048, device=dev, generator=g).bfloat16()
    return ws, x, ao


T, S = model(2048, 6144), model(1024, 3072)


def chain_today(m, keep_gu):
    ws, x, ao = m
    for wqkv, wo, wgu, wd in ws:
        ops.gemm(x, wqkv)
        ops.gemm(ao, wo, residual=x)
        if keep_gu:
            act = ops.swiglu_fwd(ops.gemm(x, wgu))
        else:
            act, _ = ops.gemm_swiglu(x, wgu, save_gu=False)
        ops.gemm(act, wd, residual=x)


def chain_budget(m, budget, splits):
    ws, x, ao = m
    so, sd = splits
    for wqkv, wo, wgu, wd in ws:
        lib.sd_debug_cu_budget(budget)
        ops.gemm_grouped_nt([(x, wqkv, 1)])
        ops.gemm_grouped_nt([(ao, wo, so)])
        act, _ = ops.gemm_grouped_nt([(x, wgu, 1)], swiglu=True)[0]
        ops.gemm_grouped_nt([(act, wd, sd)])
    lib.sd_debug_cu_budget(0)


side = ops.concurrent_stream(dev, "teacher")


def timed(fn_main, fn_side, iters=5):
    def run():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            fn_side()
        fn_main()
        main.wait_stream(side)
    run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        run()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


nothing = lambda: None
tt = timed(lambda: chain_today(T, False), nothing)
ts_ = timed(lambda: chain_today(S, True), nothing)
both = timed(lambda: chain_today(S, True), lambda: chain_today(T, False))
print(f"today: teacher alone {tt:.3f} ms, student alone {ts_:.3f} ms, two streams {both:.3f} ms (sum {tt + ts_:.3f})", flush=True)
for bt, bs in ((256, 256), (192, 64), (176, 80), (200, 56), (160, 96), (208, 48)):
    for splits_t, splits_s in (((1, 1), (1, 1)), ((1, 2), (1, 1)), ((1, 1), (1, 2))):
        a = timed(lambda: chain_budget(T, bt, splits_t), nothing)
        b = timed(lambda: chain_budget(S, bs, splits_s), nothing)
        c = timed(lambda: chain_budget(S, bs, splits_s), lambda: chain_budget(T, bt, splits_t))
        print(f"budget teacher {bt:3d} / student {bs:3d}, K slices (o,down) T{splits_t} S{splits_s}: teacher alone {a:.3f}, "
              f"student alone {b:.3f}, side by side {c:.3f} ms", flush=True)
