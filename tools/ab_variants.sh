#!/bin/bash
# usage: tools/ab_variants.sh [bench args] -- benches every gpurun_variants/lib_*.so in place of the built library (dev helper);
# prints views/s and the two blend kernels' stage times
L=saro-gs_amd/diff_gaussian_rasterization_ch3/libgsrast_hip.so
cp $L /tmp/orig.so
for v in gpurun_variants/lib_*.so; do cp $v $L; echo -n "$(basename $v): "; timeout 300 python tools/bench_brief.py --steps 200 --warmup 20 "$@" | python3 -c "
import sys, re
t = sys.stdin.read()
m = re.search(r'views/s ([0-9.]+) ms/step ([0-9.]+)', t)
g = lambda k: (re.search(r\"'%s': ([0-9.]+)\" % k, t) or [None, '?'])[1]
print('views/s', m.group(1) if m else t[:200], 'ms', m.group(2) if m else '', 'fwd', g('blend_fwd'), 'bwd', g('blend_bwd'), 'sort_depth', g('sort_depth'), 'cut_redo', g('cut_redo'), 'pbwd', g('preprocess_bwd'), 'emit', g('emit_instances'), 'sort_tile', g('sort_tile'), 'pfwd', g('preprocess_fwd'), 'lrz', g('late_rows_zero'))
"; done
cp /tmp/orig.so $L
