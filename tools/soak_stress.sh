#!/bin/bash
# usage: tools/soak_stress.sh <runs> [ENV=1 ...] -- the two-thread gate soak test <runs> times under `timeout 60` each, with the given environment (dev helper: the rare hang of DESIGN 4.4b)
n=$1; shift
hang=0
for i in $(seq 1 $n); do
  env "$@" timeout 60 python -m pytest tests/test_gpu_gate.py -m gpu -x -q -k soak > /tmp/soak_$i.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then hang=$((hang+1)); echo "run $i rc=$rc: $(tail -1 /tmp/soak_$i.log)"; fi
done
echo "soak_stress $* : $hang of $n runs failed or hung"
