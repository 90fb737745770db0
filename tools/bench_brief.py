#!/usr/bin/env python
"""Run bench.py with the given extra args and print a compact summary (development helper)."""
import json, subprocess, sys
args = sys.argv[1:]
out = subprocess.run([sys.executable, "bench.py", "--sweep", "", "--no-cpu-baseline"] + args, capture_output=True, text=True, timeout=600)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print("FAILED", args, out.stderr[-2000:]); sys.exit(1)
d = json.loads(line[-1])
print(" ".join(args), "| cold views/s", d["value"], "ms", d["ms_per_step"], "| warm", d.get("value_warm"), d.get("ms_per_step_warm"), "| pipelined", d.get("pipelined", {}).get("views_per_s_cold"),
      d.get("pipelined", {}).get("views_per_s_warm"), "| no cut", d.get("value_no_list_cut"), "| bwd", d["roofline"]["avg_launch_ms"], d.get("stage_ms"), d.get("host"))
