#!/usr/bin/env python
"""Turn a rocprofv3 --kernel-trace rocpd database (or kernel_trace CSV) into the per-kernel summary
committed under profiles/: calls, total / average / min / max duration, share of GPU time.

    python tools/rocprof_summary.py gpurun_out/prof_x/x_results.db > profiles/r01_kernel_stats.txt
"""
import csv
import sqlite3
import sys
from collections import defaultdict


def rows_from_db(path):
    cur = sqlite3.connect(path).cursor()
    for name, start, end in cur.execute("select name, start, end from kernels"):
        yield name, (end - start) / 1e3


def rows_from_csv(path):
    with open(path) as f:
        for r in csv.DictReader(f):
            yield r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3


def main():
    path = sys.argv[1]
    rows = rows_from_db(path) if path.endswith(".db") else rows_from_csv(path)
    agg = defaultdict(list)
    for name, us in rows:
        agg[name].append(us)
    tot = sum(sum(v) for v in agg.values())
    print(f"# source: {path}")
    print(f"# total kernel time: {tot / 1e3:.3f} ms over {sum(len(v) for v in agg.values())} dispatches")
    print(f"{'kernel':<72} {'calls':>6} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}")
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{name[:72]:<72} {len(v):>6} {sum(v):>12.1f} {sum(v) / len(v):>10.2f} {min(v):>10.2f} {max(v):>10.2f} {100 * sum(v) / tot:>6.2f}")


if __name__ == "__main__":
    main()
