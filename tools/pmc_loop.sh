#!/bin/bash
# usage: tools/pmc_loop.sh <tag> <kernel regex> <counter> [steady_loop options ...] -- one rocprofv3 --pmc pass over the headline loop (24 steps, pose table
# off, per-call synchronised, chain inline), per-kernel averages of the kernels that match (development helper, round 6)
tag=$1; pat=$2; ctr=$3; shift; shift; shift
out=gpurun_out/pmcl_$tag
rm -rf $out
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/$out -o pmc -- python $GRAFT_REPO_ROOT/tools/steady_loop.py 3e6 8 24 no_order_hint=1 sync=1 chain_gate=0 "$@" > /dev/null 2> /tmp/pmcl_$tag.err)
f=$(find $out -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python tools/pmc_summary.py $f | grep -E "^kernel|$pat" | cut -c1-120
rm -rf $out
