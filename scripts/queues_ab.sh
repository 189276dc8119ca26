#!/bin/bash
# A/B of the stream placement (ops.concurrent_stream) on the N>1 code path rehearsed with one rank (RCCL) and on the
# single-process path: SD_STREAM_PICK=0 takes the first streams torch hands out, GPU_MAX_HW_QUEUES widens HIP's queue pool.
run() {
  env "$@" timeout -k 10 200 python bench.py $ARGS --steps 10 --warmup 3 --no-cpu-baseline --no-prof 2>/dev/null | python -c "
import sys,json
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$ARGS $*', round(b['ms_per_step'],3), {k:round(v,2) for k,v in b['phases_ms'].items()})"
}
ARGS=--force-dist; run SD_STREAM_PICK=0; run SD_STREAM_PICK=1; run SD_STREAM_PICK=1 GPU_MAX_HW_QUEUES=8
ARGS=; run SD_STREAM_PICK=0; run SD_STREAM_PICK=1
