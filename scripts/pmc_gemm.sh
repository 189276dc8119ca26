#!/bin/bash
# Runs ON THE GPU BOX: PMC passes over tests/bench_p256.py (the gate|up / lm_head forward GEMMs on the persistent
# 256x128 kernel) to see what the K loop waits for: SQ wait / LDS / MFMA counters, then L2 hit/miss.  One counter group per pass, kernel-trace only (see the brief on --pmc).
# usage: scripts/pmc_gemm.sh <tag>;  summary: python scripts/pmc_gemm_summary.py gpurun_out/pmc_gemm_<tag>
set -e
tag=${1:-r02}
out=gpurun_out/pmc_gemm_$tag
mkdir -p $out
export TMPDIR=/tmp
export SD_GEMM_NO_P256=1
root=$(pwd)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/sq -o run -- python3 $root/tests/bench_p256.py > $out/sq.log 2>&1
echo "[pmc] sq done"
# NOT collected: a pass with TA_BUSY_avr / TA_*_STALLED_BY_TC_CYCLES_sum aborted inside rocprofv3 (signal 6) on this
# image and then sat silent until the 7-minute watchdog killed it (round 2) -- do not add TA_* counters back blindly.
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/tcc -o run -- python3 $root/tests/bench_p256.py > $out/tcc.log 2>&1
echo "[pmc] tcc done"
find $out -name "*.db" -delete
find $out -name "*kernel_trace.csv" -delete
du -sh $out
