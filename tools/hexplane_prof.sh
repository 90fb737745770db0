#!/bin/bash
# kernel-level times of the hexplane field (development helper; run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in dnerf neural3d; do
  rm -rf gpurun_out/hexprof_$w
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/hexprof_$w -o hex -- python tools/hexplane_prof.py $w "$@" > /dev/null 2>&1
  f=$(find gpurun_out/hexprof_$w -name "*kernel_stats.csv" | head -1)
  echo "== $w"; [ -n "$f" ] && head -12 "$f" | cut -d, -f1-5 | cut -c1-150
  find gpurun_out/hexprof_$w -type f ! -name "*kernel_stats.csv" -delete
done
