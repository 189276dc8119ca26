#!/usr/bin/env python3
"""usage: fetch_calibration.py <rocprofv3 --pmc FETCH_SIZE output dir of tests/calib_fetch.py> <expect.json> <out.json>"""
import json
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_summary import per_kernel, symbol  # noqa: E402

src, expect, dst = sys.argv[1], json.load(open(sys.argv[2])), sys.argv[3]
fetch = per_kernel(src, "FETCH_SIZE")
rows = {}
for raw, (n, v) in fetch.items():
    rows.setdefault(symbol(raw), [0, 0.0])
    rows[symbol(raw)][0] += n
    rows[symbol(raw)][1] += v
out = {"method": "rocprofv3 --pmc FETCH_SIZE over tests/calib_fetch.py (known-byte launches); counter unit KB; "
                 "ratio = known bytes / (FETCH_SIZE x 1024): the factor to apply to FETCH_SIZE for that load path", "kernels": {}}
for key, e in expect.items():
    hits = {k: v for k, v in rows.items() if (key in k if key != "copy" else ("copy" in k.lower() or "elementwise" in k.lower()))}
    for k, (n, v) in hits.items():
        per = v * 1024.0 / n
        out["kernels"][k[:120]] = {"what": e["what"], "known_bytes_per_launch": e["bytes_per_launch"], "launches_seen": n,
                                   "fetch_size_bytes_per_launch": per, "known_over_counter": e["bytes_per_launch"] / per if per else None}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
