#!/bin/bash
# usage: tools/ab_variants.sh -- benches every gpurun_variants/lib_*.so in place of the built library (dev helper)
L=saro-gs_amd/diff_gaussian_rasterization_ch3/libgsrast_hip.so
cp $L /tmp/orig.so
for v in gpurun_variants/lib_*.so; do cp $v $L; echo "== $v"; timeout 200 python tools/bench_brief.py --steps 30 --warmup 5 "$@" | cut -c1-260; done
cp /tmp/orig.so $L
