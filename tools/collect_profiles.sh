#!/bin/bash
# Runs on the GPU box (via gpurun): default bench, rocprofv3 kernel-trace stats of the same command,
# and the two HBM-traffic PMC passes.  Everything lands in gpurun_out/profiles_<tag>/ ; copy what is
# to be judged into profiles/ (tools/install_profiles.py does that and derives pmc_blend_bwd.json).
# usage: tools/collect_profiles.sh <tag> [bench args, e.g. --gaussians 3000000]
tag=${1:-r01}; shift
extra="$@"
out=gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python bench.py $extra > $out/bench_${tag}.json 2> $out/bench_${tag}.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python bench.py $extra --sweep "" --no-cpu-baseline > $out/bench_${tag}_under_rocprof.json 2> $out/rocprof.err
f=$(ls $out/trace/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python tools/rocprof_summary.py $f > $out/${tag}_kernel_stats.txt
s=$(ls $out/trace/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$s" ] && cp $s $out/${tag}_rocprofv3_kernel_stats.csv
rm -rf $out/trace
# (counter collection SERIALISES the device's kernels: a completion pass of the list cut behind its stream gate -- a polling wait on the caller's
# stream, the chain on another -- would never finish; the PMC passes run the chain inline, option chain_gate 0: same kernels, same work)
pmcopt="--opt chain_gate=0"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -o pmc -- python bench.py $extra $pmcopt --steps 5 --warmup 2 --sweep "" --no-cpu-baseline > /dev/null 2> $out/pmc_$c.err
  f=$(ls $out/pmc_$c/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f > $out/${tag}_pmc_$c.txt
  rm -rf $out/pmc_$c
done
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/pmc_sq -o pmc -- python bench.py $extra $pmcopt --steps 5 --warmup 2 --sweep "" --no-cpu-baseline > /dev/null 2> $out/pmc_sq.err
f=$(ls $out/pmc_sq/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$f" ] && python tools/pmc_summary.py $f > $out/${tag}_pmc_SQ.txt
rm -rf $out/pmc_sq
# L2 hit rate, fabric request sizes, LDS conflicts: one pass per group (a group a box does not know is skipped: its file stays empty)
rocprofv3 --list-avail > $out/${tag}_counters_avail.txt 2>&1
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_x$i -o pmc -- python bench.py $extra $pmcopt --steps 5 --warmup 2 --sweep "" --no-cpu-baseline > /dev/null 2> $out/pmc_x$i.err
  f=$(ls $out/pmc_x$i/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f > $out/${tag}_pmc_X$i.txt
  rm -rf $out/pmc_x$i
done
grep -c . $out/${tag}_counters_avail.txt > /dev/null && grep -o "TCC_[A-Z0-9_]*\|SQ_LDS_[A-Z_]*" $out/${tag}_counters_avail.txt | sort -u | tr '\n' ' ' > $out/${tag}_counters_tcc_lds_names.txt; rm -f $out/${tag}_counters_avail.txt
# stream timeline of one steady-state step of the headline loop (8 poses round-robin)
bash tools/timeline_run.sh ${tag}_3M 3e6 8 120 && cp gpurun_out/timeline_${tag}_3M.txt $out/${tag}_timeline_3M.txt
# the table-OFF twin (value_cold: every forward a first visit -- predicted cut depths): kernel stats and timeline of the headline loop alone
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/trace_cold -o trace -- python $GRAFT_REPO_ROOT/tools/steady_loop.py 3e6 8 400 no_order_hint=1 > /dev/null 2> $GRAFT_REPO_ROOT/$out/rocprof_cold.err; cd $GRAFT_REPO_ROOT
f=$(ls $out/trace_cold/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python tools/rocprof_summary.py $f > $out/${tag}_kernel_stats_cold.txt && python tools/timeline.py $f 5 > $out/${tag}_timeline_3M_cold.txt
rm -rf $out/trace_cold
ls -la $out
