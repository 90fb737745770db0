#!/bin/bash
# runs the GPU suite under rocgdb (batch) a few times; on SIGABRT / SIGSEGV prints the native backtraces
for i in 1 2 3 4; do
  timeout 900 /opt/rocm/bin/rocgdb -batch -ex "handle SIGUSR1 nostop noprint" -ex "handle SIG33 nostop noprint" -ex run -ex "thread apply all bt 25" -ex quit \
     --args python -m pytest tests -m gpu -q -x -p no:faulthandler > gpurun_out/gdb_$i.log 2>&1
  if grep -q "SIGABRT\|SIGSEGV\|Aborted" gpurun_out/gdb_$i.log; then echo "run $i CRASH"; grep -n "SIGABRT\|SIGSEGV" -A60 gpurun_out/gdb_$i.log | head -120; break; else echo "run $i ok: $(grep -E 'passed|failed' gpurun_out/gdb_$i.log | tail -1)"; fi
done
