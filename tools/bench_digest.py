#!/usr/bin/env python
"""Short digest of a bench.py JSON line (development helper).  usage: tools/bench_digest.py <file>"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "n_gpus", d["n_gpus"])
c = d["config"]
for k in ("poses_per_rank", "pose_table_switched_off", "list_cut_switched_off", "one_repeated_pose", "list_cut_late_gaussians", "column_runs_early", "R_eff", "instances_R"):
    print("  ", k, c.get(k))
r = d["roofline"]
print({k: r.get(k) for k in ("achieved", "frac", "avg_launch_ms", "valu_issue_slot_frac", "traffic_ratio")})
if d.get("per_stage"):
    print("per_stage", {k: (v["ms"], v["hbm_frac"]) for k, v in d["per_stage"].items() if v["ms"]})
print("step bytes", {k: v for k, v in (d.get("step_algorithmic_bytes") or {}).items() if k != "note"})
print("host", d.get("host_step_ms"))
for k in ("sweep_1080p", "two_views_in_flight_1080p"):
    if k in d:
        print(k, d[k])
if "baseline_configs" in d:
    print("cfgs", {k: v["views_per_s"] for k, v in d["baseline_configs"].items()}, "shell", d["shell_scene_1080p"]["views_per_s"])
for k, v in (d.get("training_like") or {}).items():
    print("  train", k, v)
print("eval", d.get("eval_fps_forward_only"))
if "next_rows" in d:
    print("static it", d["next_rows"].get("static_stage_training_iteration", {}).get("ms"))
print("cpu", d.get("cpu_baseline"))
