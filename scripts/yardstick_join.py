#!/usr/bin/env python3
"""profiles/<tag>_gemm_yardstick.json = tests/bench_yardstick.py's timings of record (an un-profiled run: torch.cuda events)
+ the kernel NAMES each side launched, taken from runs of the same script under `rocprofv3 --kernel-trace` (rocpd sqlite).
usage: yardstick_join.py <tag> <full yardstick.json> [<trace.db> <log of that traced run>]...
In a trace, per shape and in program order: the vendor launches (hipBLASLt, then rocBLAS: one group when both pick the
same kernel), then the in-tree launches (one group per variant tried)."""
import json
import re
import sqlite3
import sys


def groups(db):
    c = sqlite3.connect(db)
    out = []
    for name, dur, gx, wx, lds, vg in c.execute(
            "select name, duration, grid_x, workgroup_x, lds_size, vgpr_count from kernels order by start"):
        if not ("Cijk" in name or "gemm" in name.lower()):
            continue
        if out and out[-1]["name"] == name:
            out[-1]["d"].append(dur)
        else:
            out.append({"name": name, "d": [dur], "wgs": gx // max(wx, 1), "threads": wx, "lds": lds, "vgpr": vg})
    return [g for g in out if len(g["d"]) >= 4]


def brief(g):
    d = sorted(g["d"])
    n = g["name"].replace("void (anonymous namespace)::", "").split("(")[0].strip()
    return {"kernel": n[:150], "median_us_in_trace": round(d[len(d) // 2] / 1e3, 1), "launches": len(d),
            "workgroups": g["wgs"], "threads": g["threads"], "lds_bytes": g["lds"], "vgpr": g["vgpr"]}


def main():
    tag, full = sys.argv[1], json.load(open(sys.argv[2]))
    names = {}
    for db, log in zip(sys.argv[3::2], sys.argv[4::2]):
        order = [m.group(1) for m in re.finditer(r"^(c\d\.\S+)\s+(?:NT|NN|TN) ", open(log).read(), re.M)]
        gs, gi = groups(db), 0
        vendor = lambda g: g["name"].startswith(("Cijk", "Custom_Cijk"))  # noqa: E731
        for shape in order:
            v, m = [], []
            while gi < len(gs) and vendor(gs[gi]):
                v.append(brief(gs[gi])); gi += 1
            while gi < len(gs) and not vendor(gs[gi]):
                m.append(brief(gs[gi])); gi += 1
            names[shape] = {"vendor": v, "intree": m}
    for row in full["rows"]:
        row["kernels"] = names.get(row["name"])
        for k in list(row):  # the first run printed TFLOP/s with a unit slip: recompute from gflop and us
            if k.endswith("_tflops"):
                row[k] = round(row["gflop"] / row[k[:-7] + "_us"] * 1e3, 1)
    out = {"what": "vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS of ROCm 7.2) beside the in-tree kernel on every GEMM shape of "
                   "one BASELINE-config-2 step (c2.*: 2 048 tokens, lm_heads on the 1 536 loss rows) and on the config-5 shapes "
                   "(c5.*: 32 768 tokens); tests/bench_yardstick.py on one MI355X; never on the product path",
           "how": full.get("note"), "device": full.get("device"), "torch": full.get("torch"), "hip": full.get("hip"),
           "reading": "vendor_over_intree = in-tree time / best vendor time (> 1: the vendor kernel is faster); "
                      "max_rel_diff_vs_vendor 0.0 = the in-tree result equals the vendor's bit for bit; kernel names and "
                      "median_us_in_trace from separate runs under rocprofv3 --kernel-trace (profiled clocks are lower)",
           "rows": full["rows"]}
    json.dump(out, open(f"profiles/{tag}_gemm_yardstick.json", "w"), indent=1)
    for r in full["rows"]:
        k = r.get("kernels") or {}
        v = ((k.get("vendor") or [{}])[-1]).get("kernel", "?")
        i = ((k.get("intree") or [{}])[0]).get("kernel", "?")
        bv = min(r.get("hipblaslt_us", 9e9), r.get("rocblas_us", 9e9))
        bi = min(r["intree_us"], r.get("intree_splitk_us", 9e9))
        print(f"{r['name']:24s} vendor {bv:7.1f} us {r['gflop'] / bv * 1e3:6.0f} TF | in-tree {bi:7.1f} us {r['gflop'] / bi * 1e3:6.0f} TF | "
              f"x{r['vendor_over_intree']:.3f} | {re.sub(r'_MI16.*', '', v)[-46:]:46s} | {i[:52]}")


if __name__ == "__main__":
    main()
