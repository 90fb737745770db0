import os, sys, time
sys.path[:0] = ["/root/repo", "/root/repo/saro-gs_amd"]
import torch, bench, scenes
import diff_gaussian_rasterization_ch3 as rast
_C = rast._C
dev = torch.device("cuda:0")
wl = bench.Workload(rast, scenes, 3_000_000, 1920, 1080, 3, 0, 8, dev)
_C.set_option("no_order_hint", 1)
for _ in range(30): wl.step(None, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): wl.step(None, 1)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) * 10, "late", _C.context_query("last_late"), "pause", _C.context_query("cut_pause"), "early runs", _C.context_query("last_early_runs"))
_C.profile_reset(); _C.set_option("profile", -1)
for _ in range(10): wl.step(None, 1)
torch.cuda.synchronize()
pk = _C.profile_read(); _C.set_option("profile", 0)
print({k: round(v[0] / max(v[1], 1), 4) for k, v in pk.items() if v[1]})
