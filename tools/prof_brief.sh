#!/bin/bash
# usage: tools/prof_brief.sh <tag> [bench args...]  -- rocprofv3 kernel trace of a short bench run, summary to stdout
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_$tag -o $tag -- python bench.py --steps 20 --warmup 5 --sweep "" --no-cpu-baseline "$@" > gpurun_out/prof_bench_$tag.json 2>gpurun_out/prof_err_$tag.log
f=$(ls gpurun_out/prof_$tag/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then python tools/rocprof_summary.py $f > gpurun_out/prof_$tag/summary.txt; head -${LINES_OUT:-14} gpurun_out/prof_$tag/summary.txt; rm -f $f; else echo "no trace"; tail -5 gpurun_out/prof_err_$tag.log; fi
