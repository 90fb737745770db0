import sys, time
sys.path[:0] = ["/root/repo", "/root/repo/saro-gs_amd"]
import numpy as np, torch, scenes
import diff_gaussian_rasterization_ch3 as rast
_C = rast._C
dev = torch.device("cuda:0")
P, W, H = 1_000_000, 1920, 1080
sc = scenes.synth(P, 0)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
L = {k: t(sc[k]).requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
g = t(scenes.upstream_grad(H, W, 1))
cams = [scenes.camera(k, 8, W, H) for k in range(3)]
rs = [rast.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=t(sc["bg"]), scale_modifier=1.0,
      viewmatrix=t(c["viewmatrix"]), projmatrix=t(c["projmatrix"]), sh_degree=3, campos=t(c["campos"]), prefiltered=False) for c in cams]
_C.set_option("list_cut_always", 1)
rng = np.random.default_rng(1)
op0 = L["opacities"].detach().clone()
fb0 = _C.context_query("cut_fallbacks")
t0 = time.perf_counter()
N = 400
for i in range(N):
    with torch.no_grad():
        L["opacities"].copy_(op0 * float(rng.uniform(0.05, 1.0)))       # the scene turns more or less transparent between calls
    for p in list(L.values()) + [m2]:
        p.grad = None
    color, radii, depth = rast.GaussianRasterizer(rs[i % 3])(means3D=L["means3D"], means2D=m2, opacities=L["opacities"], shs=L["shs"], scales=L["scales"], rotations=L["rotations"])
    color.backward(g)
torch.cuda.synchronize()
print("ok", N, "calls", round((time.perf_counter() - t0) / N * 1e3, 3), "ms/call, completion passes", _C.context_query("cut_fallbacks") - fb0)
