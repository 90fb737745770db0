#!/bin/bash
# effect of HSA_ENABLE_INTERRUPT=0 (busy-poll instead of interrupt waits) on a short bench run
for rep in 1 2 3; do
for v in "" 0; do
  if [ -z "$v" ]; then unset HSA_ENABLE_INTERRUPT; else export HSA_ENABLE_INTERRUPT=$v; fi
  timeout 200 python bench.py --steps 20 --warmup 5 --sweep "" --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('HSA_ENABLE_INTERRUPT=${v:-unset}', d['value'], d['ms_per_step'], d['host_step_ms'])"
done; done
