"""Where is the GPU idle?  Reads a rocprofv3 --kernel-trace CSV, takes the UNION of all kernel intervals (any stream) and
reports busy time over the span, the idle time by gap size class, and the kernels that most often precede a long gap.
usage: python scripts/trace_gaps.py <kernel_trace.csv> [skip_fraction_front=0.3]"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
    rows.sort()
    rows = rows[int(len(rows) * skip):]        # drop start-up (model init, first steps)
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy, cur_end, gaps = 0, rows[0][0], []
    last_name = rows[0][2]
    for s, e, name in rows:
        if s > cur_end:
            gaps.append((s - cur_end, last_name, name))
            busy += 0
            cur_start = s
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end = e
            last_name = name
    span = t1 - t0
    print(f"kernels {len(rows)}  span {span / 1e6:.1f} ms  busy(union) {busy / 1e6:.1f} ms  idle {100 * (span - busy) / span:.2f} %")
    classes = [(0, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e6), (1e6, 1e12)]
    for lo, hi in classes:
        g = [x for x in gaps if lo <= x[0] < hi]
        print(f"gaps {lo / 1e3:6.0f}..{hi / 1e3:8.0f} us: n={len(g):6d} total {sum(x[0] for x in g) / 1e6:8.2f} ms")
    big = [x for x in gaps if x[0] >= 2e4]
    c = collections.Counter((a, b) for _, a, b in big)
    tot = collections.Counter()
    for d, a, b in big:
        tot[(a, b)] += d
    for (a, b), n in sorted(c.items(), key=lambda kv: -tot[kv[0]])[:12]:
        print(f"  {tot[(a, b)] / 1e6:7.2f} ms in {n:4d} gaps after [{a}] before [{b}]")


if __name__ == "__main__":
    main()
