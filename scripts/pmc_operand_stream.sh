#!/bin/bash
# Runs ON THE GPU BOX: PMC evidence for what paces the tile GEMMs' K loop.  Three separate
# counter passes (never combined with trace domains) over the serial bench command:
#   tcp: L1 -> L2 read requests and their summed latency (average latency of a request under load = LATENCY / REQ)
#   tcc: L2 hits / misses / requests
#   (a TA_* pass -- TA_BUSY_avr, TA_ADDR_STALLED_BY_TC_CYCLES_sum ... -- aborted rocprofv3 with signal 6 in round 2
#    (scripts/pmc_gemm.sh) and hung it until its timeout in round 3: DROPPED, not retried.  What is known: `rocprofv3 -L`
#    on this image does list the TA_* counters for gfx950 (gpurun_out/r4u/counters.txt, round 4), so it is not a missing
#    counter; the TA block has one instance per CU (256) and the `_avr` / `_sum` forms are DERIVED over all instances, while
#    ROCm 7.2 ships no gfx950 section in derived_counters.xml (MI355X_MICROARCH.md "rocprofv3 PMC slots") -- the gfx94x
#    formulas it falls back to address 304-CU instance lists.  The questions those counters were meant to answer (is the
#    address path or the data return the limiter of the operand stream?) were answered another way in round 4: three
#    load paths, one delivery ceiling (profiles/r04_fill_paths.json).)
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
