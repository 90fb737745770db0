import sys, cProfile, pstats, io
sys.path[:0] = ["/root/repo", "/root/repo/saro-gs_amd"]
import torch, bench, scenes
import diff_gaussian_rasterization_ch3 as rast
dev = torch.device("cuda:0")
wl = bench.Workload(rast, scenes, 100_000, 1920, 1080, 3, 0, 8, dev)
for i in range(50): wl.step(None, 1)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(300): wl.step(None, 1)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:4000])
