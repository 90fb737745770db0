#!/usr/bin/env python
"""List cut (include/gsrast.h: options.no_list_cut) on the bench workload: late Gaussians, fallbacks, listed instances, the per-stage
device times and the step time with the cut on and off (development helper).  usage: tools/cut_probe.py <P> [cube|shell] [W H]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "saro-gs_amd")]
import torch
import bench
import scenes
import diff_gaussian_rasterization_ch3 as rast

P = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "cube"
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
dev = torch.device("cuda:0")
_C = rast._C
wl = bench.Workload(rast, scenes, P, W, H, 3, 0, 8, dev, kind=kind)
for off in (1, 0, 1, 0):
    _C.set_option("no_list_cut", off)
    for _ in range(5):
        wl.step(None, 1)
    torch.cuda.synchronize(dev)
    fb0 = _C.context_query("cut_fallbacks")
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        wl.step(None, 1)
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / n * 1e3
    st = wl._forward_state()
    _C.profile_reset()
    _C.set_option("profile", -1)
    for _ in range(10):
        wl.step(None, 1)
    torch.cuda.synchronize(dev)
    pk = _C.profile_read()
    _C.set_option("profile", 0)
    stages = {k: round(v[0] / max(v[1], 1), 4) for k, v in pk.items() if v[1]}
    print(f"P={P} {kind} no_list_cut={off}: {ms:.4f} ms/step  late={_C.context_query('last_late')} fallbacks+={_C.context_query('cut_fallbacks') - fb0} "
          f"R={st['R']} listed={st['listed']} R_eff_listed={st['R_eff']} Q={_C.context_query('last_runs')}\n   {stages}", flush=True)
_C.set_option("no_list_cut", 0)
