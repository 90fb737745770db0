#!/usr/bin/env python
"""Where the host's time goes in ONE per-call synchronised step (bench.py's headline protocol), development helper:
   python tools/sync_probe.py [P=3e6] [poses=8] [steps=200] [option=value ...]
Wraps the two C entry points the step goes through and prints medians of: Python before the forward's C call, the forward's C call
(launches + the spin for the instance counts), Python between the two C calls (autograd), the backward's C call, and from the
backward's return to the end of torch.cuda.synchronize()."""
import statistics
import sys
import time

sys.path[:0] = ["/root/repo", "/root/repo/saro-gs_amd"]
import torch, bench, scenes  # noqa: E401,E402
import diff_gaussian_rasterization_ch3 as rast  # noqa: E402

P = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000
V = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N = int(sys.argv[3]) if len(sys.argv) > 3 else 200
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    rast._C.set_option(k, int(v))
_C = rast._C
L = _C.lib()
marks = {}


class Wrap:
    def __init__(self, fn, name):
        self.fn, self.name = fn, name

    def __call__(self, *a):
        marks[self.name + "_in"] = time.perf_counter()
        r = self.fn(*a)
        marks[self.name + "_out"] = time.perf_counter()
        return r


L.gsrast_forward_ex = Wrap(L.gsrast_forward_ex, "fwd")
L.gsrast_backward_ex = Wrap(L.gsrast_backward_ex, "bwd")
dev = torch.device("cuda:0")
wl = bench.Workload(rast, scenes, P, 1920, 1080, 3, 0, V, dev, poses=V)
rows = []
prologue = []
for i in range(N):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wl.step(None, 1)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if i >= N // 4:
        prologue.append(_C.context_query("last_prologue_ns") * 1e-9)
        rows.append((t2 - t0, marks["fwd_in"] - t0, marks["fwd_out"] - marks["fwd_in"], marks["bwd_in"] - marks["fwd_out"],
                     marks["bwd_out"] - marks["bwd_in"], t1 - marks["bwd_out"], t2 - t1))
names = ("step", "python before fwd C call", "fwd C call", "python between", "bwd C call", "python after bwd", "final sync wait")
for k, n in enumerate(names):
    print("%-28s median %8.1f us   min %8.1f" % (n, statistics.median(r[k] for r in rows) * 1e6, min(r[k] for r in rows) * 1e6))
print("%-28s median %8.1f us   min %8.1f   (inside the fwd C call: entry -> first kernel launched)" % ("C prologue", statistics.median(prologue) * 1e6, min(prologue) * 1e6))

# ---- second pass: where inside "python before fwd C call" (wrappers around the layers of the binding) ----
import diff_gaussian_rasterization_ch3 as R  # noqa: E402
stamps = {}


def stamp_on_entry(obj, attr, name):
    fn = getattr(obj, attr)

    def w(*a, **k):
        stamps[name] = time.perf_counter()
        return fn(*a, **k)
    setattr(obj, attr, w)


stamp_on_entry(R.GaussianRasterizer, "forward", "module.forward")
stamp_on_entry(R._RasterizeGaussians, "forward", "autograd.forward")
stamp_on_entry(_C, "rasterize_gaussians", "_C.rasterize_gaussians")
stamp_on_entry(_C._Arena, "acquire", "arena.acquire")
rows = []
for i in range(N):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wl.step(None, 1)
    torch.cuda.synchronize()
    if i >= N // 4:
        rows.append((stamps["module.forward"] - t0, stamps["autograd.forward"] - stamps["module.forward"], stamps["_C.rasterize_gaussians"] - stamps["autograd.forward"],
                     stamps["arena.acquire"] - stamps["_C.rasterize_gaussians"], marks["fwd_in"] - stamps["arena.acquire"]))
for k, n in enumerate(("step() -> Module.forward", "-> autograd Function.forward (apply)", "-> _C.rasterize_gaussians", "-> arena.acquire (checks, 3 torch.empty)", "-> C call (allocators, options, pointers)")):
    print("  %-44s median %6.1f us" % (n, statistics.median(r[k] for r in rows) * 1e6))
