"""Turns the rocprofv3 --pmc passes of scripts/profile_round.sh into profiles/<tag>_pmc_traffic.json (mode `traffic`:
every kernel symbol, with the GB/s it sustained) and profiles/<tag>_mfma_busy.json (mode `mfma`).  Legacy default mode:
HBM bytes per launch of the dominant GEMM kernel (and of every NT GEMM), per MI355X_MICROARCH.md section HBM:
counter unit KB, FETCH_SIZE doubled on gfx950 for 16-B/lane streaming reads.
usage: python scripts/pmc_summary.py gpurun_out/prof_r01 profiles/r01_pmc_traffic.json"""
import csv
import glob
import json
import re
import sys


def per_kernel(path, counter):
    acc = {}
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            a = acc.setdefault(row["Kernel_Name"], [0, 0.0])
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return acc


def mfma_busy(src, dst, n_cu=256):
    """MFMA-busy share of each kernel's active time: SQ_VALU_MFMA_BUSY_CYCLES (cycles summed over the 4 SIMDs of every
    CU) / (active cycles x CUs x 4) -- the gfx94x MfmaUtil formula (ROCm 7.2 has no gfx950 derived counters).
    rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs, so active cycles = GRBM_GUI_ACTIVE / 8 (checked against
    the event-timed TFLOP/s of the GEMM kernels: 0.48 busy for the kernel that runs at 1 145 of 2 500 TFLOP/s)."""
    busy = per_kernel(src + "/mfma", "SQ_VALU_MFMA_BUSY_CYCLES")
    act = per_kernel(src + "/mfma", "GRBM_GUI_ACTIVE")
    rows = []
    tb = ta = 0.0
    for k, (n, b) in busy.items():
        a = act.get(k, [0, 0.0])[1]
        if a <= 0:
            continue
        tb += b
        ta += a
        if b > 0:
            rows.append({"kernel": k[:160], "launches": n, "gui_active_cycles": a, "mfma_busy_share": b / (a / 8.0 * n_cu * 4)})
    rows.sort(key=lambda r: -r["gui_active_cycles"])
    out = {"method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over `python3 bench.py --steps 2 --warmup 1 "
                     "--no-cpu-baseline --no-prof --no-overlap`; share = busy / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)",
           "whole_run_mfma_busy_share": tb / (ta / 8.0 * n_cu * 4) if ta else None, "kernels": rows[:40]}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps({"whole_run": out["whole_run_mfma_busy_share"], "top": [(r["kernel"][:70], round(r["mfma_busy_share"], 3)) for r in rows[:12]]}, indent=1))


def symbol(name):
    """rocprofv3 kernel name -> the symbol bench.py reports: demangled, without `void`, namespace and argument list."""
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if m:  # rocprofv3 leaves names with __bf16 template arguments mangled (no demangler here knows DF16b): decode the
        # simple forms our kernels use -- <name>I(Li<int>E | Lb<0|1>E | DF16b | f)*E
        n = int(m.group(1))
        base, rest = name[m.end():m.end() + n], name[m.end() + n:]
        args = []
        if rest.startswith("I"):
            rest = rest[1:]
            while rest and not rest.startswith("E"):
                a = re.match(r"Li(\d+)E|Lb([01])E|(DF16b)|(f)", rest)
                if not a:
                    args = None
                    break
                args.append(a.group(1) if a.group(1) else ("true" if a.group(2) == "1" else "false") if a.group(2)
                            else "__bf16" if a.group(3) else "float")
                rest = rest[a.end():]
        return base + ("<" + ", ".join(args) + ">" if args else "")
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(name):  # cut at the '(' that opens the argument list (outside template brackets)
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def traffic_all(src, dst):
    """Per kernel symbol: HBM-side bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, KB units) and, with the average
    duration of the same kernel in the serial --kernel-trace pass, the HBM-side GB/s it sustained."""
    fetch = per_kernel(src + "/fetch", "FETCH_SIZE")
    write = per_kernel(src + "/write", "WRITE_SIZE")
    dur = {}
    for f in glob.glob(src + "/serial/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            dur[symbol(row["Name"])] = (float(row["AverageNs"]) / 1e3, int(row["Calls"]))
    agg = {}
    for raw, (n, v) in fetch.items():
        a = agg.setdefault(symbol(raw), {"launches": 0, "fetch": 0.0, "wl": 0, "write": 0.0})
        a["launches"] += n
        a["fetch"] += v * 1024.0 * 2.0
    for raw, (n, v) in write.items():
        a = agg.setdefault(symbol(raw), {"launches": 0, "fetch": 0.0, "wl": 0, "write": 0.0})
        a["wl"] += n
        a["write"] += v * 1024.0
    kernels = {}
    for sym, a in agg.items():
        if not a["launches"] or not a["wl"] or sym.startswith(("at::", "rocprim", "__amd")):
            continue
        fb, wb = a["fetch"] / a["launches"], a["write"] / a["wl"]
        e = {"launches": a["launches"], "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
             "hbm_bytes_per_launch": fb + wb}
        if sym in dur:
            e["avg_us_serial_trace"] = dur[sym][0]
            e["hbm_gbs"] = (fb + wb) / dur[sym][0] / 1e3
        kernels[sym] = e
    out = {"method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `python3 bench.py --steps 2 "
                     "--warmup 1 --no-cpu-baseline --no-prof --no-overlap` (scripts/profile_round.sh); counter unit KB; "
                     "FETCH_SIZE doubled (gfx950 reports half the bytes of 16-B/lane streaming reads, "
                     "MI355X_MICROARCH.md section HBM; other access widths are uncalibrated); per-launch averages over "
                     "all launches of the kernel; avg_us_serial_trace / hbm_gbs use the same kernel's average duration "
                     "in the --kernel-trace --stats pass of the same command (profiles/<tag>_bench_serial_kernel_stats.csv). "
                     "These are memory-side (fabric) bytes: Infinity-Cache hits are counted, not excluded",
           "kernels": dict(sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))}
    json.dump(out, open(dst, "w"), indent=1)
    for k in list(out["kernels"])[:25]:
        e = out["kernels"][k]
        print(f"{k[:60]:60s} n={e['launches']:5d} {e['hbm_bytes_per_launch'] / 1e6:9.2f} MB/launch  {e.get('avg_us_serial_trace', 0):8.1f} us  {e.get('hbm_gbs', 0):7.0f} GB/s")


def main():
    if len(sys.argv) > 3 and sys.argv[3] == "mfma":
        return mfma_busy(sys.argv[1], sys.argv[2])
    if len(sys.argv) > 3 and sys.argv[3] == "traffic":
        return traffic_all(sys.argv[1], sys.argv[2])
    src, dst = sys.argv[1], sys.argv[2]
    fetch = per_kernel(src + "/fetch", "FETCH_SIZE")
    write = per_kernel(src + "/write", "WRITE_SIZE")

    def summarise(pattern, label):
        rx = re.compile(pattern)
        n = sum(v[0] for k, v in fetch.items() if rx.search(k))
        fb = sum(v[1] for k, v in fetch.items() if rx.search(k)) * 1024.0 * 2.0
        nw = sum(v[0] for k, v in write.items() if rx.search(k))
        wb = sum(v[1] for k, v in write.items() if rx.search(k)) * 1024.0
        if not n or not nw:
            return None
        return {"kernel": label, "launches": n, "fetch_bytes_per_launch": fb / n, "write_bytes_per_launch": wb / nw,
                "hbm_bytes_per_launch": fb / n + wb / nw}
    out = {"method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `python3 bench.py --steps 2 "
                     "--warmup 1 --no-cpu-baseline --no-prof --no-overlap` (scripts/profile_round.sh); counter unit KB; "
                     "FETCH_SIZE doubled (gfx950 reports half the bytes of 16-B/lane streaming reads, "
                     "MI355X_MICROARCH.md section HBM); per-launch averages over all launches of the named kernel(s)",
           "gemm_nt_stag": summarise(r"gemm_stag_kernel<false, false, \d>|gemm_pstag_kernel<4, false, false, \d>",
                                     "gemm_pstag_kernel<4,false,false,EPI> + gemm_stag_kernel<false,false,EPI> (the staggered 256x128 forward family)"),
           "nt_gemm": summarise(r"gemm_(bf16|stag|ks|pstag)_kernel<(\d+, \d+, |\d+, )?false, false",
                                "every NT GEMM launch: gemm_bf16_kernel<*,*,false,false,*> + gemm_stag_kernel<false,false,*>")}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
