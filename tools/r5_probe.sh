#!/bin/bash
# round-5 scratch: headline + cold stage tables, dynamic legs
O=gpurun_out/$1; mkdir -p $O
python tools/bench_brief.py --steps 200 --warmup 20 > $O/brief_warm.txt 2>&1
python tools/bench_brief.py --steps 200 --warmup 20 --opt no_order_hint=1 > $O/brief_cold.txt 2>&1
python tools/bench_brief.py --steps 200 --warmup 20 --opt no_order_hint=1 --opt tau_cut=0 > $O/brief_cold_notau.txt 2>&1
python tools/bench_brief.py --steps 200 --warmup 20 --gaussians 1000000 > $O/brief_warm_1M.txt 2>&1
python tools/bench_brief.py --steps 200 --warmup 20 --gaussians 1000000 --opt no_order_hint=1 > $O/brief_cold_1M.txt 2>&1
python tools/bench_brief.py --steps 200 --warmup 20 --gaussians 300000 > $O/brief_warm_300k.txt 2>&1
python tools/bench_brief.py --steps 200 --warmup 20 --gaussians 300000 --opt late_fill_min_p=0 > $O/brief_warm_300k_fill.txt 2>&1
GSRAST_TRACE=1 LEGS=dynamic_opacity,dynamic_full python tools/pose_cycle_probe.py train 3e6 8 > $O/dyn.txt 2> $O/dyn_trace.txt
for f in $O/brief_*.txt $O/dyn.txt; do echo "== $f"; cut -c1-1100 $f; done
grep -c "completion pass" $O/dyn_trace.txt
