#!/bin/bash
# Runs on the GPU box (via gpurun).  Round 6 form: the bench line (+ the side legs' report), rocprofv3 kernel-trace stats of the same
# bench command, and -- on the HEADLINE LOOP alone (tools/steady_loop.py 3e6 8 N no_order_hint=1 sync=1: pose table off, every call
# synchronised, exactly what bench.py times) -- kernel stats, stream timeline and the hardware-counter passes.  Everything lands in
# gpurun_out/profiles_<tag>/ ; tools/install_profiles.py copies the set into profiles/ and derives pmc_blend_{bwd,fwd}_3M.json.
# usage: tools/collect_profiles.sh <tag> [bench args]
tag=${1:-r06}; shift
extra="$@"
out=gpurun_out/profiles_$tag
mkdir -p $out
loop="3e6 8 240 no_order_hint=1 sync=1"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout 900 python bench.py $extra --extras $out/bench_report_${tag}.json > $out/bench_${tag}_with_extras.out 2> $out/bench_${tag}.err ) 2> $out/bench_${tag}_with_extras.time
tail -1 $out/bench_${tag}_with_extras.out > $out/bench_${tag}.json
( time timeout 600 python bench.py $extra --steps 20 --warmup 5 > $out/bench_${tag}_driver_command.json 2>> $out/bench_${tag}.err ) 2> $out/bench_${tag}_driver_command.time
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python bench.py $extra --sweep "" --no-cpu-baseline > $out/bench_${tag}_under_rocprof.json 2> $out/rocprof.err
f=$(ls $out/trace/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python tools/rocprof_summary.py $f > $out/${tag}_kernel_stats_bench_command.txt
s=$(ls $out/trace/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$s" ] && cp $s $out/${tag}_rocprofv3_kernel_stats.csv
rm -rf $out/trace
# the headline loop alone: kernel stats + timeline (pose table off, per-call synchronised) and its table-on twin
for twin in cold warm; do
  args="$loop"; [ $twin = warm ] && args="3e6 8 240 sync=1"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/trace_$twin -o trace -- python $GRAFT_REPO_ROOT/tools/steady_loop.py $args > $GRAFT_REPO_ROOT/$out/steady_$twin.log 2> $GRAFT_REPO_ROOT/$out/rocprof_$twin.err)
  f=$(ls $out/trace_$twin/*kernel_trace.csv 2>/dev/null | head -1)
  suffix=""; [ $twin = warm ] && suffix="_warm"
  [ -n "$f" ] && python tools/rocprof_summary.py $f > $out/${tag}_kernel_stats${suffix}.txt && python tools/timeline.py $f 5 > $out/${tag}_timeline_3M${suffix}.txt
  rm -rf $out/trace_$twin
done
# (counter collection SERIALISES the device's kernels: a completion pass of the list cut behind its stream gate would never finish; the
# library sees ROCPROF_COUNTERS in its environment and runs the chain inline -- chain_gate=0 is passed as well, belt and braces)
pmc() {   # name, counters...
  name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_$name -o pmc -- python $GRAFT_REPO_ROOT/tools/steady_loop.py 3e6 8 24 no_order_hint=1 sync=1 chain_gate=0 > /dev/null 2> $GRAFT_REPO_ROOT/$out/pmc_$name.err)
  f=$(ls $out/pmc_$name/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f > $out/${tag}_pmc_$name.txt
  rm -rf $out/pmc_$name
}
pmc FETCH_SIZE FETCH_SIZE
pmc WRITE_SIZE WRITE_SIZE
pmc SQ SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
pmc X1 TCC_HIT_sum TCC_MISS_sum
pmc X2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pmc X3 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pmc X4 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
python tools/sync_probe.py 3e6 8 200 no_order_hint=1 > $out/${tag}_sync_probe.txt 2>&1
ls -la $out
