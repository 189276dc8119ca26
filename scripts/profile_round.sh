#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the rocprofv3 evidence behind bench.py's `roofline` object.
#   1. --kernel-trace --stats of the default bench command (two overlapped streams)
#   2. the same with --no-prof --no-overlap (one stream: per-kernel durations comparable with the JSON `kernels` table)
#   3. --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (never combined with other trace domains)
#   4. --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE: MFMA-busy per kernel (scripts/pmc_summary.py -> <tag>_mfma_busy.json)
# Output under gpurun_out/prof_<tag>/ ; scripts/pmc_summary.py turns (3) into profiles/<tag>_pmc_traffic.json.
# usage: scripts/profile_round.sh r01
set -e
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/default -o run -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $out/default.log 2>&1
echo "[profile] default done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/serial -o run -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof --no-overlap --no-traffic > $out/serial.log 2>&1
echo "[profile] serial done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o run -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-overlap --no-traffic > $out/fetch.log 2>&1
echo "[profile] fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o run -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-overlap --no-traffic > $out/write.log 2>&1
echo "[profile] write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/mfma -o run -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-overlap --no-traffic > $out/mfma.log 2>&1
echo "[profile] mfma done"
find $out -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | sort
# keep the merge-back under gpurun's 64 MiB: traces are not needed, only the stats / counter tables
find $out -name "*kernel_trace.csv" -delete
find $out -name "*.db" -delete
du -sh $out
