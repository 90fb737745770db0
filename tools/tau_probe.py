#!/usr/bin/env python
"""Can a cheap optical-depth estimate predict which tiles saturate (development probe)?  tau of an 8 x 8 pixel block = sum over the
Gaussians whose centre falls into it of opacity * 2 pi sqrt(det cov2D) / 64; a tile is PREDICTED to saturate iff all four of its
blocks have tau >= tau0.  Compared with the tiles that really saturate (every pixel's n_contrib short of the list's end)."""
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
sub = int(sys.argv[4]) if len(sys.argv) > 4 else 16          # sampling: every sub-th Gaussian
W, H = 1920, 1080
dev = torch.device("cuda:0")
sc = scenes.synth(P, 0) if kind == "cube" else scenes.synth_shell(P, 0)
cam = scenes.camera(k, 8, W, H)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
e = torch.empty(0)
_C = rast._C
_C.set_option("no_list_cut", 1)
R, color, radii, gb, bb, ib, depth = _C.rasterize_gaussians(t(sc["bg"]), t(sc["means3D"]), e, t(sc["opacities"]), t(sc["scales"]), t(sc["rotations"]), 1.0, e,
    t(cam["viewmatrix"]), t(cam["projmatrix"]), cam["tanfovx"], cam["tanfovy"], H, W, t(sc["shs"]), 3, t(cam["campos"]), False)
st = _C.debug_export(P, R, W, H, gb, bb, ib)
gy, gx = (H + 15) // 16, (W + 15) // 16
T = gx * gy
nc = st["n_contrib"].to(torch.int64)
rg = st["ranges"].to(torch.int64); ln = (rg[:, 1] - rg[:, 0])
pad = torch.full((gy * 16, gx * 16), -1, dtype=torch.int64, device=dev); pad[:H, :W] = nc
lnpix = ln.view(gy, 1, gx, 1).expand(gy, 16, gx, 16).reshape(gy * 16, gx * 16)
# a pixel inside the image is saturated iff it stopped before the end of its tile's list (conservative: last contributor < list length)
inside = pad >= 0
ft = torch.zeros((gy * 16, gx * 16), device=dev); ft[:H, :W] = st["final_T"].view(H, W)
sat_pix = (~inside) | (ft < 3e-4)          # approximately: final_T is the transmittance BEFORE the entry that ended the pixel
sat_tile = sat_pix.view(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(T, 256).all(dim=1)
tm = torch.where(inside, pad, torch.zeros_like(pad)).view(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(T, 256).amax(dim=1)
vis = radii > 0
co = st["conic_opacity"]; m2 = st["means2D"]
det_conic = co[:, 0] * co[:, 2] - co[:, 1] ** 2
area = 2 * np.pi / torch.sqrt(det_conic.clamp_min(1e-12))        # 2 pi sqrt(det cov) = 2 pi / sqrt(det conic)
w = co[:, 3] * area.clamp_max(256.0) / 64.0
sel = vis & (torch.arange(P, device=dev) % sub == 0)
bx = (m2[:, 0] / 8).floor().long().clamp(0, gx * 2 - 1); by = (m2[:, 1] / 8).floor().long().clamp(0, gy * 2 - 1)
tau = torch.zeros(gy * 2 * gx * 2, device=dev).index_add_(0, (by * gx * 2 + bx)[sel], w[sel] * sub)
tau_tile = tau.view(gy, 2, gx, 2).permute(0, 2, 1, 3).reshape(T, 4).amin(dim=1)
print(f"{kind} P={P} pose {k} sampling 1/{sub}: tiles {T}, really saturating {int(sat_tile.sum())}; consumed by non-saturating tiles: {int(tm[~sat_tile].sum())} of {int(tm.sum())}")
for tau0 in (10, 20, 40, 80, 160):
    pred = tau_tile >= tau0
    fp = pred & ~sat_tile            # predicted to saturate, does not: the completion pass blends it again (its consumed depth = a serial chain)
    fn = ~pred & sat_tile            # saturates, not predicted: listed in full for nothing
    print(f"  tau0 {tau0:4d}: predicted {int(pred.sum())}; false positives {int(fp.sum())} (deepest consumed {int(tm[fp].max()) if fp.any() else 0}, sum {int(tm[fp].sum())}); "
          f"missed {int(fn.sum())} (listed {int(ln[fn].sum())} of {int(ln.sum())})")
