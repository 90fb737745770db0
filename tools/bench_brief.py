#!/usr/bin/env python
"""Run bench.py with the given extra args and print a compact summary (development helper)."""
import json, subprocess, sys
args = sys.argv[1:]
out = subprocess.run([sys.executable, "bench.py", "--sweep", "", "--no-cpu-baseline", "--no-training-like"] + args, capture_output=True, text=True, timeout=600)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print("FAILED", args, out.stderr[-2000:]); sys.exit(1)
d = json.loads(line[-1])
ps = {k: round(v["ms"], 3) if isinstance(v, dict) and "ms" in v else v for k, v in d.get("per_stage", {}).items()}
print(" ".join(args), "| views/s", d["value"], "ms/step", d["ms_per_step"], ps or d["kernels_ms"], d.get("host_step_ms"))
