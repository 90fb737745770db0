#!/usr/bin/env python
"""Development helper: per-kernel microseconds per fwd+bwd pass from tools/hexplane_prof.sh's stats (6 passes each)."""
import csv
for w in ("dnerf", "neural3d"):
    rows = list(csv.reader(open(f"gpurun_out/hexprof_{w}/hex_kernel_stats.csv")))[1:]
    tot = 0.0
    print("==", w)
    for r in rows:
        per_pass = float(r[2]) / 6 / 1e3
        if per_pass > 3:
            print(f"  {r[0][:60]:60s} calls/pass {int(r[1]) / 6:5.1f}  us/pass {per_pass:8.1f}")
        tot += per_pass
    print(f"  total GPU us/pass {tot:.1f}")
