#!/bin/bash
# Runs ON THE GPU BOX: PMC evidence for what paces the tile GEMMs' K loop.  Three separate
# counter passes (never combined with trace domains) over the serial bench command:
#   tcp: L1 -> L2 read requests and their summed latency (average latency of a request under load = LATENCY / REQ)
#   tcc: L2 hits / misses / requests
#   (a TA_* pass -- TA_BUSY_avr, TA_ADDR_STALLED_BY_TC_CYCLES_sum ... -- hangs rocprofv3 on this pool until its timeout
#    kills it, in round 2 and again in round 3: not collected)
# scripts/pmc_operand_summary.py turns them into profiles/<tag>_pmc_operand_stream.json.
set -u
tag=${1:-r03}
out=gpurun_out/pmc_stream_$tag
mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
cmd="python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-overlap"
timeout -k 10 240 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE --output-format csv -d $out/tcp -o run -- $cmd > $out/tcp.log 2>&1; echo "[pmc] tcp rc=$?"
timeout -k 10 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/tcc -o run -- $cmd > $out/tcc.log 2>&1; echo "[pmc] tcc rc=$?"
find $out -name "*.db" -delete
find $out -name "*counter_collection.csv" | sort
du -sh $out
