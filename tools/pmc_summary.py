#!/usr/bin/env python
"""Per-kernel average of rocprofv3 --pmc counters (counter_collection.csv) -> text table.
A kernel's PREDICATED launches (the list cut's completion pass: `if (*pred == 0) return`) count nothing; they are left out of the
averages: a dispatch counts if its first counter reaches 5 % of the kernel's largest dispatch (`calls` = dispatches that count, `all` = all)."""
import csv, sys
from collections import defaultdict
per = defaultdict(lambda: defaultdict(dict))          # kernel -> dispatch -> counter -> value
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        per[r["Kernel_Name"][:60]][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({c for d in per.values() for v in d.values() for c in v})
print(f"{'kernel':<62}{'calls':>6} " + " ".join(f"{n:>22}" for n in names) + f"{'all':>8}")
rows = []
for k, disp in per.items():
    key = names[0]
    top = max(v.get(key, 0.0) for v in disp.values())
    keep = [v for v in disp.values() if v.get(key, 0.0) >= 0.05 * top] or list(disp.values())
    avg = {c: sum(v.get(c, 0.0) for v in keep) / len(keep) for c in names}
    rows.append((sum(avg.values()) * len(keep), k, len(keep), avg, len(disp)))
for _, k, n, avg, nall in sorted(rows, reverse=True):
    print(f"{k:<62}{n:>6} " + " ".join(f"{avg[c]:>22.4g}" for c in names) + f"{nall:>8}")
