#!/bin/bash
# usage: tools/pmc_brief.sh <tag> "<counters>" [bench args]   -- one rocprofv3 --pmc pass (kernel-trace only), per-kernel sums
tag=$1; shift; ctrs=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/pmc_$tag -o $tag -- python bench.py --steps 3 --warmup 1 --sweep "" --no-cpu-baseline "$@" > gpurun_out/pmc_bench_$tag.json 2>gpurun_out/pmc_err_$tag.log
f=$(ls gpurun_out/pmc_$tag/*counter_collection.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then python tools/pmc_summary.py $f > gpurun_out/pmc_$tag/summary.txt; cat gpurun_out/pmc_$tag/summary.txt; rm -f $f gpurun_out/pmc_$tag/*kernel_trace.csv; else echo "no counters"; tail -5 gpurun_out/pmc_err_$tag.log; fi
