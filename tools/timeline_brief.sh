#!/bin/bash
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$tag -o $tag -- python bench.py --steps 30 --warmup 5 --sweep "" --no-cpu-baseline "$@" > gpurun_out/tl_bench_$tag.json 2>/dev/null
f=$(ls gpurun_out/tl_$tag/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python tools/step_timeline.py $f 20 && python tools/step_timeline.py $f 25 | tail -1 && rm -f $f
