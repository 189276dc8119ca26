# A/B of bench.py launch modes (overlap on/off, teacher serial, cached loss rows, full head): prints ms/step and phases per mode, twice
set -e
cd $GRAFT_REPO_ROOT
for mode in "" "--serial-teacher" "--no-overlap" "--experiment-cached-rows" "--full-head"; do
  for rep in 1 2; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-prof $mode 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('MODE[$mode] rep$rep ms=%.3f phases=%s' % (d['ms_per_step'], {k:round(v,3) for k,v in d['phases_ms'].items()}))"
  done
done
