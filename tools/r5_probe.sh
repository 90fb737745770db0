#!/bin/bash
# round-5 scratch: headline + cold stage tables of the current library and of round 4's (gpurun_variants/lib_r4.so), dynamic-leg trace
L=saro-gs_amd/diff_gaussian_rasterization_ch3/libgsrast_hip.so
O=gpurun_out/$1; mkdir -p $O
python tools/bench_brief.py --steps 200 --warmup 20 > $O/brief_warm.txt 2>&1
python tools/bench_brief.py --steps 200 --warmup 20 --opt no_order_hint=1 > $O/brief_cold.txt 2>&1
python tools/bench_brief.py --steps 200 --warmup 20 --gaussians 1000000 > $O/brief_warm_1M.txt 2>&1
python tools/bench_brief.py --steps 200 --warmup 20 --gaussians 1000000 --opt late_fill_min_p=0 > $O/brief_warm_1M_fill.txt 2>&1
if [ -f gpurun_variants/lib_r4.so ]; then
  cp $L /tmp/cur.so; cp gpurun_variants/lib_r4.so $L
  python tools/bench_brief.py --steps 200 --warmup 20 > $O/brief_warm_r4.txt 2>&1
  python tools/bench_brief.py --steps 200 --warmup 20 --opt no_order_hint=1 > $O/brief_cold_r4.txt 2>&1
  cp /tmp/cur.so $L
fi
GSRAST_TRACE=1 LEGS=dynamic_opacity python tools/pose_cycle_probe.py train 3e6 8 > $O/dyn.txt 2> $O/dyn_trace.txt
for f in $O/brief_*.txt $O/dyn.txt; do echo "== $f"; cut -c1-1500 $f; done
grep -c "" $O/dyn_trace.txt
