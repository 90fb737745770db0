#!/bin/bash
# first GPU process on a fresh box: does the bench show the slow-host anomaly?  prints value + per-step host diagnostics
P=${1:-1000000}
BENCH_STEP_TRACE=1 timeout 300 python bench.py --steps 30 --warmup 5 --gaussians $P --sweep "" --no-cpu-baseline 2> gpurun_out/first_run.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('FIRST', d['value'], d['ms_per_step'], d['host_step_ms'])"
grep "step host" gpurun_out/first_run.err | cut -c1-400
timeout 300 python bench.py --steps 30 --warmup 5 --gaussians $P --sweep "" --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('SECOND', d['value'], d['ms_per_step'], d['host_step_ms'])"
