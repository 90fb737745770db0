#!/bin/bash
# usage: tools/timeline.sh <tag> [bench args...] -- rocprofv3 kernel trace of a short bench run -> gpurun_out/timeline_<tag>.txt (+ per-kernel summary)
# The step shown is the 75th from the end: behind the headline's 20 timed steps come two more timed runs of 25 steps (launch-order hints off, list cut off), 3 steps back on the default path, the 2
# forward-only calls of the statistics and the 10 event-bracketed steps of the stage table (whose event records put ~10 us of idle queue in front of every stage).
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$tag -o $tag -- python bench.py --steps 20 --warmup 5 --sweep "" --no-cpu-baseline "$@" > gpurun_out/tl_bench_$tag.json 2>gpurun_out/tl_err_$tag.log
f=$(ls gpurun_out/tl_$tag/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then
  python tools/timeline.py $f ${TL_BACK:-75} > gpurun_out/timeline_$tag.txt
  python tools/rocprof_summary.py $f > gpurun_out/kstats_$tag.txt
  head -3 $f > gpurun_out/tl_header_$tag.txt
  rm -rf gpurun_out/tl_$tag
  tail -25 gpurun_out/timeline_$tag.txt
else echo "no trace"; tail -5 gpurun_out/tl_err_$tag.log; fi
