#!/bin/bash
# usage: tools/ab_sync.sh <rounds> <kernel-regex or -> [steady_loop args ...]
# Round 6's A/B: every gpurun_variants/lib_*.so in place of the built library, `rounds` alternating passes of the headline loop
# (tools/steady_loop.py 3e6 8 300 no_order_hint=1 sync=1 = bench.py's protocol: pose table off, per-call synchronised), then -- unless the
# regex is "-" -- one rocprofv3 kernel-trace pass per variant with the average time of the kernels that match.
rounds=${1:-2}; pat=${2:--}; shift; shift
args="${@:-3e6 8 300 no_order_hint=1 sync=1}"
L=saro-gs_amd/diff_gaussian_rasterization_ch3/libgsrast_hip.so
cp $L /tmp/orig.so
for r in $(seq $rounds); do
  for v in gpurun_variants/lib_*.so; do
    cp $v $L
    echo "$(basename $v .so) round $r: $(timeout 300 python tools/steady_loop.py $args 2>&1 | grep -E 'per-call|steady_loop' | sed 's/steady_loop P=[0-9]* poses=[0-9]*//' | tr '\n' ' ' | cut -c1-200)"
  done
done
if [ "$pat" != "-" ]; then
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  for v in gpurun_variants/lib_*.so; do
    cp $v $L; tag=$(basename $v .so)
    rm -rf gpurun_out/abp_$tag
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abp_$tag -o t -- python $GRAFT_REPO_ROOT/tools/steady_loop.py $args > /dev/null 2>&1)
    f=$(find gpurun_out/abp_$tag -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" "$pat" "$tag" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
for r in rows:
    if pat.search(r["Name"]):
        print(f"{sys.argv[3]:12s} {r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs']) / 1e3:8.2f}")
PY
    rm -rf gpurun_out/abp_$tag
  done
fi
cp /tmp/orig.so $L
