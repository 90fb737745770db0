#!/bin/bash
# repeated short bench runs right after a pytest process, with per-step host times (looking for the rare slow-host run)
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tiny or ragged" 2>&1 | tail -1
for i in 1 2 3 4 5 6; do
  BENCH_STEP_TRACE=1 GSRAST_TRACE=${GT:-} timeout 200 python bench.py --steps 20 --warmup 5 --sweep "" --no-cpu-baseline 2> gpurun_out/anom_$i.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('run $i', d['value'], d['ms_per_step'])"
  grep "step host" gpurun_out/anom_$i.err | cut -c1-200
done
