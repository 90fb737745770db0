#!/usr/bin/env python
"""Per-kernel average of rocprofv3 --pmc counters (counter_collection.csv) -> text table."""
import csv, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float)); calls = defaultdict(set)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
names = sorted({c for v in agg.values() for c in v})
print(f"{'kernel':<62}{'calls':>6} " + " ".join(f"{n:>22}" for n in names))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    n = len(calls[k])
    print(f"{k:<62}{n:>6} " + " ".join(f"{v.get(c, 0) / n:>22.4g}" for c in names))
