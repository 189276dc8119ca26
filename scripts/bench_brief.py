"""Print the headline fields of a bench.py JSON line (file given as argv[1])."""
import json
import sys

b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(b["ms_per_step"], 3), "ms", round(b["value"]), "tok/s", {k: round(v, 2) for k, v in b["phases_ms"].items()},
      "optimizer", b.get("optimizer_step_ms"))
for k, v in sorted(b.get("kernels", {}).items()):
    print(f"  {k:14s} {v['ms_per_step']:7.3f} ms/step  {v['launches_per_step']:5.0f} x {v['avg_us']:7.1f} us  "
          + (f"{v['tflops']:7.0f} TF/s" if "tflops" in v else f"{v['gbs']:7.0f} GB/s"))
