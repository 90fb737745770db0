#!/usr/bin/env python
"""Short digest of a bench.py JSON line or of a bench_report.json (development helper).  usage: tools/bench_digest.py <file>"""
import json, sys
txt = open(sys.argv[1]).read()
try:
    d = json.loads(txt)
except ValueError:
    d = json.loads([l for l in txt.splitlines() if l.startswith("{")][-1])
if "headline" in d:            # a report of tools/bench_extras.py
    for k, v in d.items():
        if k in ("per_stage",) and v:
            print("per_stage", {n: (s["ms"], s["hbm_frac"]) for n, s in v.items() if s["ms"]})
        elif k == "training_like" and isinstance(v, dict):
            for n, row in v.items():
                print("  train", n, row)
        elif k in ("sweep_1M_1080p", "shell_scene_1080p") and isinstance(v, dict):
            print(k, {n: v.get(n) for n in ("views_per_s", "ms_per_step")})
        else:
            print(k, v if len(str(v)) < 600 else str(v)[:600] + " ...")
    sys.exit(0)
print("value (cold)", d["value"], "ms", d["ms_per_step"], "| warm", d.get("value_warm"), d.get("ms_per_step_warm"), "| n_gpus", d["n_gpus"], "| pipelined", d.get("pipelined"))
print("config", d["config"])
for k in ("roofline", "roofline_fwd"):
    r = d.get(k) or {}
    print(k, {n: r.get(n) for n in ("achieved", "frac", "avg_launch_ms", "valu_issue_slot_frac", "valu_issue_slot_frac_guide_2cyc_fma", "traffic_ratio")})
print("stage_ms", d.get("stage_ms"))
print("sweep", d.get("sweep_1080p_cold"), "no cut", d.get("value_no_list_cut"))
print("cpu", d.get("cpu_baseline"))
