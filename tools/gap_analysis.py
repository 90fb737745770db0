#!/usr/bin/env python
"""GPU idle time between consecutive kernels in a rocprofv3 kernel_trace.csv (steady-state part)."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50]))
rows.sort()
rows = rows[len(rows) // 3:]          # skip warm-up
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = {}
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = s1 - e0
    if g > 0:
        k = f"{n0[:28]} -> {n1[:28]}"
        a = gaps.setdefault(k, [0, 0]); a[0] += g; a[1] += 1
print(f"span {span/1e6:.3f} ms, busy {busy/1e6:.3f} ms ({100*busy/span:.1f}%), idle {(span-busy)/1e6:.3f} ms")
for k, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"{g/1e3:9.1f} us total  {g/n/1e3:7.2f} us avg x{n:4d}  {k}")
