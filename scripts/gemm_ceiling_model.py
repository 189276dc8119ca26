#!/usr/bin/env python3
"""The arithmetic behind DESIGN.md section 8 "Formal restatement of the target" (round 4).  No GPU needed.

Model: a CU takes in at most BW = 66 GB/s of operand bytes under the GEMM's panel sharing (profiles/r04_fill_paths.json: the
same through LDS-DMA, global->VGPR and global->VGPR->LDS) and the matrix pipes give 2.5 PFLOP/s x (2.05 / 2.4 GHz held under
load).  A GEMM [M,N,K] on tiles a x b takes ceil(tiles / 256) rounds of max(delivery, MFMA) time per tile; split-K into s
slices multiplies the tiles and pays s fp32 slabs written and read once at 5 TB/s.  Zero prologue, epilogue and launch cost.
For every GEMM of a BASELINE-config-2 micro-step the best (tile, s) is taken -- once with a 256x256 tile allowed, once with
tiles up to 256x128 (what 12 waves of 168 registers hold; the in-tree dispatch)."""
import math

BW, MFMA, HBM = 66e9, 2.5e15 * 2.05 / 2.4, 5.0e12
M, R, V = 2048, 1536, 159488


def best(m, n, k, splits, tiles):
    out = None
    for a, b in tiles:
        for s in splits:
            rounds = math.ceil(math.ceil(m / a) * math.ceil(n / b) * s / 256)
            t = max(rounds * (a + b) * (k / s) * 2 / BW, rounds * 2 * a * b * (k / s) / (MFMA / 256))
            if s > 1:
                t += s * m * n * 4 * 2 / HBM
            if out is None or t < out[0]:
                out = (t, a, b, s, rounds)
    return out


def run(tiles, verbose=True):
    tot = {}

    def add(group, name, m, n, k, count, splits=(1,)):
        t, a, b, s, r = best(m, n, k, splits, tiles)
        fl = 2.0 * m * n * k
        tot.setdefault(group, [0.0, 0.0])
        tot[group][0] += t * count
        tot[group][1] += fl * count
        if verbose:
            print(f"  {group:12s} {name:11s} x{count:2d}: tile {a}x{b} split {s} rounds {r:2d} {t * 1e6:7.1f} us {fl / t / 1e12:6.0f} TFLOP/s")
    sp = (1, 2, 4, 8)
    for h, inter, grp in ((1024, 3072, "student_fwd"), (2048, 6144, "teacher_fwd")):
        add(grp, "qkv", M, 4096, h, 28)
        add(grp, "o", M, h, 2048, 28, sp)
        add(grp, "gate|up", M, 2 * inter, h, 28)
        add(grp, "down", M, h, inter, 28, sp)
        add(grp, "lm_head", R, V, h, 1)
    h, inter = 1024, 3072
    add("student_bwd", "down dX", M, inter, h, 28, sp)
    add("student_bwd", "gate|up dX", M, h, 2 * inter, 28, sp)
    add("student_bwd", "o dX", M, 2048, h, 28, sp)
    add("student_bwd", "qkv dX", M, h, 4096, 28, sp)
    # a layer's four weight gradients as one persistent launch: 480 tiles of 256x128 = 2 rounds, K = 2048 tokens
    t = 2 * (256 + 128) * 2048 * 2 / BW
    fl = 2.0 * 2048 * (4096 * 1024 + 6144 * 1024 + 1024 * 3072 + 1024 * 2048)
    tot["student_bwd"][0] += t * 28
    tot["student_bwd"][1] += fl * 28
    add("student_bwd", "lm_head dX", R, h, V, 1, sp)
    add("student_bwd", "lm_head dW", V, h, R, 1)
    return tot


if __name__ == "__main__":
    for label, tiles in (("a 256x256 tile allowed", ((256, 256), (256, 128), (128, 128), (64, 128))),
                         ("tiles up to 256x128", ((256, 128), (128, 128), (64, 128)))):
        print("==", label)
        tot = run(tiles)
        at = af = 0.0
        for g, (t, f) in tot.items():
            print(f"  -> {g:12s} ideal GEMM time {t * 1e3:6.2f} ms for {f / 1e12:5.2f} TFLOP = {f / t / 2.5e15:5.3f} of the 2.5 PFLOP/s peak")
            at += t
            af += f
        print(f"  -> step GEMMs {at * 1e3:6.2f} ms = {af / at / 2.5e15:5.3f} of peak")
