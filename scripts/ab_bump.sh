# A/B of the gemm.fwd_bump experiment (sd_hip_debug.h) under the default (overlapped) bench and with the teacher serial
cd $GRAFT_REPO_ROOT
for bump in 0 1 2 3 0; do
  for mode in "" "--serial-teacher"; do
    SD_DEBUG="gemm.fwd_bump=$bump" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-prof $mode 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bump=$bump MODE[$mode] ms=%.3f phases=%s' % (d['ms_per_step'], {k:round(v,3) for k,v in d['phases_ms'].items()}))"
  done
done
