#!/usr/bin/env python
"""cProfile of the host side of the per-call synchronised step (development helper): python tools/host_profile.py [P=100000] [fwd]
`fwd`: the forward alone under torch.no_grad() (what runs in front of the first launch)."""
import sys, cProfile, pstats, io
sys.path[:0] = ["/root/repo", "/root/repo/saro-gs_amd"]
import torch, bench, scenes
import diff_gaussian_rasterization_ch3 as rast
P = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000
dev = torch.device("cuda:0")
wl = bench.Workload(rast, scenes, P, 1920, 1080, 3, 0, 8, dev, poses=8)
for i in range(50): wl.step(None, 1)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(300):
    wl.step(None, 1)
    torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:7000])
