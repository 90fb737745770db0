#!/usr/bin/env python
"""A camera PATH: N poses on a fine ring (step degrees apart), each rendered ONCE -- every call is a pose the table has never seen
(rendering a test trajectory / a video).  With option near_pose the forward borrows the previous frame's launch order and cut depths.
   python tools/path_probe.py [P=3e6] [frames=90] [ring=120 (3 degrees)] [fwdbwd=0]"""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "saro-gs_amd"))
import diff_gaussian_rasterization_ch3 as rast
import scenes
P = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 90
V = int(sys.argv[3]) if len(sys.argv) > 3 else 120
BWD = int(sys.argv[4]) if len(sys.argv) > 4 else 0
W, H, deg = 1920, 1080, 3
dev = torch.device("cuda:0")
_C = rast._C
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
sc = scenes.synth(P, 0, sh_degree=deg)
bg = t(sc["bg"])
L = {k: t(sc[k]).requires_grad_(bool(BWD)) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
m2 = torch.zeros((P, 3), device=dev, requires_grad=bool(BWD))
g = t(scenes.upstream_grad(H, W, 1))


def settings(k):
    cam = scenes.camera(k, V, W, H)
    return rast.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg, scale_modifier=1.0,
                                              viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=deg, campos=t(cam["campos"]), prefiltered=False)


def run(first, opts):
    for k_, v_ in opts.items():
        _C.set_option(k_, v_)
    rs = [settings(first + i) for i in range(N)]
    torch.cuda.synchronize()
    lates, fb0 = [], int(_C.context_query("cut_fallbacks"))
    t0 = time.perf_counter()
    for r in rs:
        for p in list(L.values()) + [m2]:
            p.grad = None
        ctx = torch.enable_grad() if BWD else torch.no_grad()
        with ctx:
            color, radii, depth = rast.GaussianRasterizer(r)(means3D=L["means3D"], means2D=m2, opacities=L["opacities"], shs=L["shs"], scales=L["scales"], rotations=L["rotations"])
            if BWD:
                color.backward(g)
        lates.append(int(_C.context_query("last_late")))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for k_ in opts:
        _C.set_option(k_, 0 if k_ != "near_pose" else 3)
    return dict(ms_per_frame=round(dt / N * 1e3, 4), fps=round(N / dt, 1), mean_late=int(np.mean(lates)), frames_cut=int(np.sum(np.array(lates) > 0)),
                completions=int(_C.context_query("cut_fallbacks")) - fb0, pause=int(_C.context_query("cut_pause")))


# warm the context on poses far from the path (allocations, capacity hints), then DISJOINT stretches of the ring per mode
assert 3 * N + 8 <= V, "three disjoint stretches of N frames + the warm-up poses must fit the ring"
for k in (V - 3, V - 2, V - 1):
    run(k, {})
out = dict(P=P, frames=N, step_deg=360.0 / V, fwdbwd=BWD)
for rad in (3, 1, 5):
    pass
out["no_list_cut"] = run(0, {"no_list_cut": 1})
out["own_slot_only"] = run(N, {"near_pose": 0})
NP = int(os.environ.get("NEAR", "3"))
_C.set_option("near_pose", NP)
out["near_pose_%d" % NP] = run(2 * N, {})
print(json.dumps(out))
