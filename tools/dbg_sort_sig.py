"""dev helper: print the depth sort's device-side verdict (significant bits, base) of one forward at P Gaussians."""
import sys, os
sys.path.insert(0, "saro-gs_amd"); sys.path.insert(0, ".")
import numpy as np, torch
import diff_gaussian_rasterization_ch3 as rast, scenes
from diff_gaussian_rasterization_ch3 import _C
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
W, H = 1920, 1080
dev = torch.device("cuda:0")
sc = scenes.synth(P, 0); cam = scenes.camera(0, 1, W, H)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
e = torch.empty(0)
R, color, radii, gb, bb, ib, depth = _C.rasterize_gaussians(t(sc["bg"]), t(sc["means3D"]), e, t(sc["opacities"]), t(sc["scales"]), t(sc["rotations"]), 1.0, e,
    t(cam["viewmatrix"]), t(cam["projmatrix"]), cam["tanfovx"], cam["tanfovy"], H, W, t(sc["shs"]), 3, t(cam["campos"]), False)
al = lambda x: (x + 255) & ~255
nblk = (P + 2047) // 2048
off = gb.numel() - 256 - al(2 * nblk * 4) - 2 * al(4 * P) - al(64 * P) - 256
sc_words = gb[off: off + 64].view(torch.int32).cpu().numpy().view(np.uint32)
st = _C.debug_export(P, R, W, H, gb, bb, ib)
d = st["depths"][radii > 0]
print("scalars", [hex(int(x)) for x in sc_words[:12]], "depth range", float(d.min()), float(d.max()))
