#!/bin/bash
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap_$tag -o $tag -- python bench.py --steps 30 --warmup 5 --sweep "" --no-cpu-baseline "$@" > gpurun_out/gap_bench_$tag.json 2>/dev/null
f=$(ls gpurun_out/gap_$tag/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python tools/gap_analysis.py $f && rm -f $f
