#!/usr/bin/env python
"""One steady-state step from a rocprofv3 kernel_trace.csv: every kernel with its duration and the idle gap before it."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "preprocess_fwd_kernel" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
a, b = starts[k], starts[k + 1]
prev_end = rows[a - 1][1]
tot_busy = tot_gap = 0
for s, e, n in rows[a:b]:
    n = n.replace("gsrast::", "").replace("void ", "")[:44]
    print(f"{(s - prev_end) / 1e3:8.2f} us gap  {(e - s) / 1e3:8.2f} us  {n}")
    tot_busy += e - s; tot_gap += s - prev_end; prev_end = e
print(f"step: busy {tot_busy / 1e3:.1f} us, gaps {tot_gap / 1e3:.1f} us, kernels {b - a}")
