"""Per-kernel averages of the counters collected by scripts/pmc_gemm.sh.  usage: python scripts/pmc_gemm_summary.py <dir> [out.json]"""
import csv
import glob
import json
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_summary import symbol  # noqa: E402


def main():
    src = sys.argv[1]
    acc = {}
    for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = symbol(row["Kernel_Name"])
            if not k.startswith("gemm_"):
                continue
            a = acc.setdefault(k, {}).setdefault(row["Counter_Name"], [0, 0.0])
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    out = {k: {c: v[1] / v[0] for c, v in cs.items()} for k, cs in acc.items()}
    for k, cs in out.items():
        if "SQ_WAVE_CYCLES" in cs:
            wc = cs["SQ_WAVE_CYCLES"]
            cs["derived"] = {"wait_any_frac": cs.get("SQ_WAIT_ANY", 0) / wc, "wait_inst_any_frac": cs.get("SQ_WAIT_INST_ANY", 0) / wc,
                             "active_inst_frac": cs.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                             "wait_inst_lds_frac": cs.get("SQ_WAIT_INST_LDS", 0) / wc,
                             "lds_conflict_per_active": cs.get("SQ_LDS_BANK_CONFLICT", 0) / max(cs.get("SQ_LDS_IDX_ACTIVE", 1), 1)}
        if "TCC_HIT_sum" in cs:
            cs.setdefault("derived", {})["l2_hit_rate"] = cs["TCC_HIT_sum"] / max(cs["TCC_HIT_sum"] + cs["TCC_MISS_sum"], 1)
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
