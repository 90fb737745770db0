#!/bin/bash
# SQ counters of the hexplane kernels (development helper; run on the GPU box): tools/hexplane_pmc.sh [dnerf|neural3d]
w=${1:-dnerf}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/hexpmc_$w
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/hexpmc_$w -o hex -- python tools/hexplane_prof.py $w > /dev/null 2>&1
f=$(find gpurun_out/hexpmc_$w -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python tools/pmc_summary.py $f | grep "kernel\|hex_" | cut -c1-260
find gpurun_out/hexpmc_$w -type f -delete
