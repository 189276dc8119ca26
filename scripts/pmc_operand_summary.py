"""profiles/<tag>_pmc_operand_stream.json from scripts/pmc_operand_stream.sh: per kernel symbol, L1 -> L2 read requests per
launch, their average latency under load (TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ, in cycles of GRBM's clock), the L2
hit rate, and -- with the kernel's duration in the serial trace of the same command -- the request rate per CU.
usage: python scripts/pmc_operand_summary.py gpurun_out/pmc_stream_r03 profiles/r03_bench_serial_kernel_stats.csv profiles/r03_pmc_operand_stream.json"""
import csv
import json
import sys

from pmc_summary import per_kernel, symbol


def main():
    src, stats, dst = sys.argv[1:4]
    dur = {}
    for row in csv.DictReader(open(stats)):
        dur[symbol(row["Name"])] = float(row["AverageNs"]) / 1e3
    agg = {}
    for sub, names in (("tcp", ("TCP_TCC_READ_REQ_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "GRBM_GUI_ACTIVE")),
                       ("tcc", ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"))):
        for name in names:
            for raw, (n, v) in per_kernel(src + "/" + sub, name).items():
                a = agg.setdefault(symbol(raw), {})
                a[name] = a.get(name, 0.0) + v
                a["launches_" + sub] = max(a.get("launches_" + sub, 0), n)
    out = {}
    for sym, a in agg.items():
        n = a.get("launches_tcp", 0)
        req = a.get("TCP_TCC_READ_REQ_sum", 0.0)
        if not n or req <= 0 or sym.startswith(("at::", "rocprim", "__amd")):
            continue
        e = {"launches": n, "l2_read_requests_per_launch": req / n,
             "avg_read_latency_cycles": a.get("TCP_TCC_READ_REQ_LATENCY_sum", 0.0) / req}
        hit, miss = a.get("TCC_HIT_sum", 0.0), a.get("TCC_MISS_sum", 0.0)
        if hit + miss > 0:
            e["l2_hit_rate"] = hit / (hit + miss)
        if sym in dur:
            e["avg_us_serial_trace"] = dur[sym]
            for line in (64, 128):  # the request size is not documented for gfx950: both readings are given
                e[f"l2_to_cu_GBps_per_cu_if_{line}B_requests"] = req / n * line / (dur[sym] * 1e-6) / 1e9 / 256
        out[sym] = e
    out = dict(sorted(out.items(), key=lambda kv: -kv[1]["l2_read_requests_per_launch"] * kv[1]["launches"]))
    json.dump({"method": "rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE | TCC_HIT_sum "
                         "TCC_MISS_sum TCC_REQ_sum (separate passes) over `python3 bench.py --steps 2 --warmup 1 "
                         "--no-cpu-baseline --no-prof --no-overlap`; a TA_* pass hangs rocprofv3 on this pool (killed by "
                         "its timeout) and is not collected", "kernels": out}, open(dst, "w"), indent=1)
    for k in list(out)[:16]:
        e = out[k]
        print(f"{k[:52]:52s} n={e['launches']:4d} req/launch {e['l2_read_requests_per_launch'] / 1e6:8.2f} M  lat {e['avg_read_latency_cycles']:7.0f} cyc"
              f"  hit {e.get('l2_hit_rate', float('nan')):.2f}  {e.get('avg_us_serial_trace', 0):7.1f} us"
              f"  {e.get('l2_to_cu_GBps_per_cu_if_64B_requests', 0):6.1f} | {e.get('l2_to_cu_GBps_per_cu_if_128B_requests', 0):6.1f} GB/s/CU")


if __name__ == "__main__":
    main()
