#!/usr/bin/env python
"""Stream timeline of ONE steady-state step from a rocprofv3 --kernel-trace CSV: for every dispatch of the step its queue,
start offset, duration and the idle gap on its own queue in front of it; then the step's critical-path summary
(sum of kernel time on the main queue, sum of gaps, step length).

    python tools/timeline.py <kernel_trace.csv> [step_index_from_end=3] > profiles/r03_timeline.txt

A step is delimited by consecutive launches of preprocess_fwd_kernel (the first kernel of a forward on the caller's stream).
"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "preprocess_fwd_kernel" in r[2] or "preprocess_raw_fwd_kernel" in r[2]]
    if len(starts) < back + 2:
        print("too few steps in the trace"); return
    # steady-state statistics over the last `n` steps except the very last one
    lens = [(rows[starts[k + 1]][0] - rows[starts[k]][0]) / 1e3 for k in range(len(starts) - 1)]
    a, b = starts[-back - 1], starts[-back]
    t0 = rows[a][0]
    step = rows[a:b]
    main_q = rows[a][3]
    print(f"# source: {path}")
    print(f"# step {len(starts) - back - 1} of {len(starts)}: {len(step)} dispatches, {(rows[b][0] - t0) / 1e3:.1f} us start-to-start "
          f"(median over all steps {sorted(lens)[len(lens) // 2]:.1f} us)")
    print(f"{'start_us':>9} {'dur_us':>8} {'gap_us':>7} {'queue':>6} kernel")
    last_end = defaultdict(lambda: None)
    ksum = defaultdict(float)
    gsum = defaultdict(float)
    gaps = []
    for s, e, name, q, st in step:
        gap = (s - last_end[q]) / 1e3 if last_end[q] is not None else 0.0
        short = name.split("(")[0].replace("void ", "").replace("gsrast::", "")[:60]
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f} {q:>6} {short}")
        ksum[q] += (e - s) / 1e3
        if last_end[q] is not None and gap > 0:
            gsum[q] += gap
            gaps.append((gap, short, q))
        last_end[q] = max(e, last_end[q] or 0)
    print("# per queue: kernel time / idle gaps between its dispatches (us)")
    for q in ksum:
        print(f"#   queue {q}{' (caller stream)' if q == main_q else ''}: {ksum[q]:.1f} / {gsum[q]:.1f}")
    print("# largest gaps (us, in front of):")
    for g, n, q in sorted(gaps, reverse=True)[:12]:
        print(f"#   {g:7.1f}  {n}  [queue {q}]")


if __name__ == "__main__":
    main()
