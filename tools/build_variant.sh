#!/bin/bash
# usage: tools/build_variant.sh <name> [-DFOO=1 ...] -- compiles gpurun_variants/lib_<name>.so with extra flags (dev helper; A/B runs via tools/ab_variants.sh)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p gpurun_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -munsafe-fp-atomics -fno-slp-vectorize -Wall -Wno-unused-function "$@" saro-gs_amd/csrc/gsrast_capi.hip -o gpurun_variants/lib_$name.so
echo gpurun_variants/lib_$name.so
