#!/bin/bash
# usage: tl.sh tag [steady_loop args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_$tag -o tl -- python $GRAFT_REPO_ROOT/tools/steady_loop.py "$@" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/tl_$tag -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f 5 > gpurun_out/timeline_$tag.txt
rm -rf gpurun_out/tl_$tag
