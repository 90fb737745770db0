#!/bin/bash
# usage: tools/ab_variants_prof.sh <kernel-name-regex> [bench args] -- rocprofv3 kernel times of the matching kernels under every
# gpurun_variants/lib_*.so (dev helper, runs on the GPU box)
pat=$1; shift
L=saro-gs_amd/diff_gaussian_rasterization_ch3/libgsrast_hip.so
cp $L /tmp/orig.so
for v in gpurun_variants/lib_*.so; do
  cp $v $L; tag=$(basename $v .so)
  LINES_OUT=40 bash tools/prof_brief.sh $tag "$@" | grep -E "$pat|total kernel" | cut -c1-140 | sed "s|^|$tag: |"
done
cp /tmp/orig.so $L
