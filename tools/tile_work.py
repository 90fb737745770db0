#!/usr/bin/env python
"""Distribution of the per-tile work of the blend kernels (development helper, runs on the GPU box):
listed instances per tile (ranges), deepest consumed list position per tile (tile_max = the backward's walk, the forward's early exit)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "saro-gs_amd"))
import diff_gaussian_rasterization_ch3 as rast
import scenes
P = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "cube"
W, H = 1920, 1080
dev = torch.device("cuda:0")
sc = scenes.synth(P, 0) if kind == "cube" else scenes.synth_shell(P, 0)
cam = scenes.camera(0, 1, W, H)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
e = torch.empty(0)
_C = rast._C
R, color, radii, gb, bb, ib, depth = _C.rasterize_gaussians(t(sc["bg"]), t(sc["means3D"]), e, t(sc["opacities"]), t(sc["scales"]), t(sc["rotations"]), 1.0, e,
    t(cam["viewmatrix"]), t(cam["projmatrix"]), cam["tanfovx"], cam["tanfovy"], H, W, t(sc["shs"]), 3, t(cam["campos"]), False)
st = _C.debug_export(P, R, W, H, gb, bb, ib)
nc = st["n_contrib"].to(torch.int64)
gy, gx = (H + 15) // 16, (W + 15) // 16
pad = torch.zeros((gy * 16, gx * 16), dtype=torch.int64, device=dev); pad[:H, :W] = nc
tm = pad.view(gy, 16, gx, 16).amax(dim=(1, 3)).flatten().cpu().numpy()
rg = st["ranges"].to(torch.int64).cpu().numpy(); ln = rg[:, 1] - rg[:, 0]
for name, a in (("listed per tile", ln), ("tile_max (consumed depth)", tm)):
    q = np.percentile(a, [50, 90, 99, 99.9, 100])
    print(f"{kind} P={P} {name}: sum {a.sum()} mean {a.mean():.1f} p50 {q[0]:.0f} p90 {q[1]:.0f} p99 {q[2]:.0f} p99.9 {q[3]:.0f} max {q[4]:.0f}; tiles > 1024: {(a > 1024).sum()}, > 4096: {(a > 4096).sum()}")
top = np.argsort(-tm)[:8]
print("heaviest tiles (x, y, tile_max, listed):", [(int(i % gx), int(i // gx), int(tm[i]), int(ln[i])) for i in top])
