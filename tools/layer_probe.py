#!/usr/bin/env python
"""What a STATELESS two-pass forward would see (development probe): pass 1 lists only the nearest fraction f of the Gaussians (by depth
rank), pass 2 completes the tiles whose pixels did not all saturate inside their pass-1 list.  Per f: tiles left for pass 2, the
instances pass 1 lists, the Gaussians (beyond the layer) whose rectangle touches a tile left, and the instances pass 2 lists.
usage: tools/layer_probe.py <P> [cube|shell] [pose k of 8]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "saro-gs_amd"))
import diff_gaussian_rasterization_ch3 as rast
import scenes
P = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "cube"
k = int(sys.argv[3]) if len(sys.argv) > 3 else 0
W, H = 1920, 1080
dev = torch.device("cuda:0")
sc = scenes.synth(P, 0) if kind == "cube" else scenes.synth_shell(P, 0)
cam = scenes.camera(k, 8, W, H)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
e = torch.empty(0)
_C = rast._C
_C.set_option("no_order_hint", 1)
R, color, radii, gb, bb, ib, depth = _C.rasterize_gaussians(t(sc["bg"]), t(sc["means3D"]), e, t(sc["opacities"]), t(sc["scales"]), t(sc["rotations"]), 1.0, e,
    t(cam["viewmatrix"]), t(cam["projmatrix"]), cam["tanfovx"], cam["tanfovy"], H, W, t(sc["shs"]), 3, t(cam["campos"]), False)
st = _C.debug_export(P, R, W, H, gb, bb, ib)
gy, gx = (H + 15) // 16, (W + 15) // 16
T = gx * gy
nc = st["n_contrib"].to(torch.int64)
fT = st["final_T"]
pad = torch.zeros((gy * 16, gx * 16), dtype=torch.int64, device=dev); pad[:H, :W] = nc
tm = pad.view(gy, 16, gx, 16).amax(dim=(1, 3)).flatten()
rg = st["ranges"].to(torch.int64)
ln = rg[:, 1] - rg[:, 0]
pl = st["point_list"].to(torch.int64)[: int(ln.sum())]
dep = st["depths"]
vis = radii > 0
rank = torch.empty(P, dtype=torch.int64, device=dev)
order = torch.argsort(torch.where(vis, dep, torch.full_like(dep, 1e30)))
rank[order] = torch.arange(P, device=dev)
nvis = int(vis.sum())
tile_of = torch.repeat_interleave(torch.arange(T, device=dev), ln)
# a pixel is saturated iff it terminated: its n_contrib < list length is not enough; use: the tile's pixels all stopped before the list's end
# (conservative proxy: tile_max < listed length, and the list is long enough to saturate: checked against pass-1 length below)
print(f"{kind} P={P} pose {k}: visible {nvis}, listed {int(ln.sum())}, consumed (sum tile_max) {int(tm.sum())}")
for f in (1 / 16, 1 / 8, 1 / 4):
    lim = int(nvis * f)
    in1 = rank[pl] < lim
    n1 = torch.zeros(T, dtype=torch.int64, device=dev).index_add_(0, tile_of, in1.to(torch.int64))
    # done after pass 1: every pixel's consumption lies inside the pass-1 prefix AND the tile saturated (its pixels stopped short of the FULL list)
    done = (tm < n1) & (tm < ln)
    left = ~done
    inst1 = int(n1.sum())
    inst2 = int((ln - n1)[left].sum())
    # Gaussians beyond the layer listed in a tile left
    g2 = torch.unique(pl[(~in1) & left[tile_of]]).numel()
    print(f"  layer = nearest {f:.4f}: pass 1 lists {inst1} instances ({inst1 / max(int(ln.sum()), 1):.3f}); tiles left {int(left.sum())} of {T}; "
          f"pass 2 lists {inst2 + int(n1[left].sum())} instances over {g2} late Gaussians (+ the layer's in those tiles)")
