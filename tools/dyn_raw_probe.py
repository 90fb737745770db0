#!/usr/bin/env python
"""Development helper: the raw entry points with all four residuals (the dynamic stage's call shape), forward + backward, for rocprofv3."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "saro-gs_amd"))
import diff_gaussian_rasterization_ch3 as rast, scenes
P, W, H = 1_000_000, 1920, 1080
dev = torch.device("cuda:0")
sc = scenes.synth(P, 0); cam = scenes.camera(0, 1, W, H)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
rs = rast.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=t(sc["bg"]), scale_modifier=1.0,
    viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=3, campos=t(cam["campos"]), prefiltered=False)
raw = dict(xyz=t(sc["means3D"]), rotation=t(sc["rotations"]), scaling=torch.log(t(sc["scales"])), opacity=torch.logit(t(sc["opacities"]).clamp(1e-4, 1 - 1e-4)),
           f_dc=t(sc["shs"][:, :1]), f_rest=t(sc["shs"][:, 1:]))
raw = {k: v.requires_grad_(True) for k, v in raw.items()}
g_ = torch.Generator(device="cpu").manual_seed(3)
res = dict(motion_residual=(0.01 * torch.randn((P, 3), generator=g_)).to(dev).requires_grad_(True), rot_residual=(0.05 * torch.randn((P, 7), generator=g_)).to(dev).requires_grad_(True),
           trbfoutput=torch.rand((P, 1), generator=g_).to(dev).requires_grad_(True), shs_residual=(0.03 * torch.randn((P, 16, 3), generator=g_)).to(dev).requires_grad_(True))
use_res = len(sys.argv) < 2 or sys.argv[1] != "nores"
gcol = torch.randn((3, H, W), generator=g_).to(dev) / (3.0 * H * W)
m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
R = rast.GaussianRasterizerRaw(rs)
def step():
    for v in list(raw.values()) + list(res.values()) + [m2]:
        v.grad = None
    c, _, _ = R(raw["xyz"], m2, raw["rotation"], raw["scaling"], raw["opacity"], raw["f_dc"], raw["f_rest"], **(res if use_res else {}))
    c.backward(gcol)
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): step()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / 30 * 1e3)
if len(sys.argv) > 1 and sys.argv[1] == "static":
    # the static-stage training iteration of bench.py's next_rows (raw rasterizer -> loss -> backward -> Adam), for a kernel trace
    import fused_adam, fused_loss
    gt = torch.rand(3, H, W, device=dev)
    inv = torch.ones(P, 1, device=dev)
    lr = dict(xyz=1.6e-4, f_dc=2.5e-3, f_rest=1.25e-4, opacity=5e-2, scaling=5e-3, rotation=1e-3)
    opt = fused_adam.GaussianAdam([{"params": [raw[k]], "lr": lr[k] * inv if k != "f_rest" else lr[k], "name": k} for k in raw], eps=1e-15)
    def it():
        color, _, _ = R(raw["xyz"], m2, raw["rotation"], raw["scaling"], raw["opacity"], raw["f_dc"], raw["f_rest"])
        loss = fused_loss.l1_dssim_loss(color, gt, 0.2)
        opt.zero_grad(); m2.grad = None
        loss.backward()
        opt.step()
    for _ in range(5): it()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): it()
    torch.cuda.synchronize(); print("static iteration ms", (time.perf_counter() - t0) / 30 * 1e3)
