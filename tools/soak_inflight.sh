#!/bin/bash
# usage: tools/soak_inflight.sh <runs> [ENV=1 ...] -- tests/test_gpu_multirank.py::test_distributed_step_with_two_views_in_flight_on_the_real_chain (one thread, two
# streams, two contexts: two forwards' gate waits and side-stream waits in flight at once) <runs> times under `timeout 120` each (dev helper)
n=$1; shift
hang=0
for i in $(seq 1 $n); do
  env "$@" timeout 120 python -m pytest tests/test_gpu_multirank.py -m gpu -x -q -k "two_views_in_flight" > /tmp/soakif_$i.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then hang=$((hang+1)); echo "run $i rc=$rc: $(tail -1 /tmp/soakif_$i.log)"; fi
done
echo "soak_inflight $* : $hang of $n runs failed or hung"
