cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/touchpmc
for tb in 1 0; do
 for c in WRITE_SIZE FETCH_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/touchpmc/p_${c}_$tb -o pmc -- python tools/steady_loop.py 3e6 8 24 chain_gate=0 touch_bits=$tb > /dev/null 2> gpurun_out/touchpmc/err_${c}_$tb.txt
  f=$(ls gpurun_out/touchpmc/p_${c}_$tb/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f > gpurun_out/touchpmc/${c}_touch$tb.txt
  rm -rf gpurun_out/touchpmc/p_${c}_$tb
 done
done
