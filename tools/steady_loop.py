#!/usr/bin/env python
"""The headline loop and nothing else (development helper for rocprofv3 --kernel-trace + tools/timeline.py):
   python tools/steady_loop.py [P=3e6] [poses=8] [steps=160] [option=value ...] [sync=1]
   sync=1: every step followed by a device synchronisation (bench.py's headline protocol); prints the median too."""
import sys
sys.path[:0] = ["/root/repo", "/root/repo/saro-gs_amd"]
import torch, bench, scenes
import diff_gaussian_rasterization_ch3 as rast
P = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000
V = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N = int(sys.argv[3]) if len(sys.argv) > 3 else 160
SYNC = False
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    if k == "sync":
        SYNC = bool(int(v))
        continue
    rast._C.set_option(k, int(v))
dev = torch.device("cuda:0")
wl = bench.Workload(rast, scenes, P, 1920, 1080, 3, 0, V, dev, poses=V)      # V poses of the ring dealt round-robin, as the headline
for i in range(N // 2):
    wl.step(None, 1)
    if SYNC:
        torch.cuda.synchronize()
torch.cuda.synchronize()
import time
ts = []
t0 = time.perf_counter()
for i in range(N - N // 2):
    s0 = time.perf_counter()
    wl.step(None, 1)
    if SYNC:
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - s0)
torch.cuda.synchronize()
if ts:
    import statistics
    print("per-call synchronised: median %.4f ms, min %.4f, max %.4f" % (statistics.median(ts) * 1e3, min(ts) * 1e3, max(ts) * 1e3))
print("steady_loop P=%d poses=%d %s: %.4f ms per step (contexts with streams %d, overlapping calls seen %d)" % (P, V, " ".join(sys.argv[4:]), (time.perf_counter() - t0) * 1e3 / (N - N // 2),
      rast._C.get_option("stream_contexts"), rast._C.get_option("concurrent_callers")))
