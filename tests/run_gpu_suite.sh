#!/bin/bash
# Runs GPU steps one after another; stops at the first step that hit its timeout (rc 124/137).
# usage: tests/run_gpu_suite.sh "<name>|<timeout s>|<command>" ...
mkdir -p gpurun_out
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; tmo="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== [$name] $cmd"
  timeout -k 10 "$tmo" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== [$name] rc=$rc"; tail -n 25 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== [$name] TIMED OUT: stopping"; exit 1; fi
done
exit 0
