#!/usr/bin/env python
"""Dev helper: one configuration of tests/test_gpu_fuzz.py, gradient errors of HIP and of the fp32 oracle against the fp64 oracle,
with both depth sorts.  usage: tools/fuzz_one.py <seed> [large]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "saro-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import scenes, diff_gaussian_rasterization_ch3 as rast
from oracle import oracle as orc
from gpu_harness import run_hip
import test_gpu_fuzz as F
seed = int(sys.argv[1]); large = len(sys.argv) > 2
orc.build()
c = F._config(seed, large=large); rng = c["rng"]; P, W, H = c["P"], c["W"], c["H"]
sc = scenes.synth(P, 2000 + seed, sh_degree=c["deg"], scale_mul=c["scale_mul"])
if c["aniso"] != 1.0:
    sc["scales"][:, 0] *= (c["aniso"] ** rng.uniform(0, 1, P)).astype(np.float32)
sc["opacities"] = (sc["opacities"] * c["opac_mul"]).astype(np.float32); sc["bg"] = c["bg"]
cam = scenes.camera(c["cam"][0], c["cam"][1], W, H); cam["scale_modifier"] = c["scale_modifier"]
g = scenes.upstream_grad(H, W, 3000 + seed) * ((H * W) if c["big_grad"] else 1.0)
o32 = orc.render(sc, cam, g); o64 = orc.render(sc, cam, g, f64=True)
for ds in (0, 1):
    rast._C.set_option("depth_sort", ds)
    h = run_hip(rast, sc, cam, torch.device("cuda:0"), dL_dcolor=g, tile_clip=1)
    print("depth_sort", ds, "R", h["R"], o32["R"], "color equal", np.array_equal(h["out_color"].view(np.uint32), o32["out_color"].view(np.uint32)))
    for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dscales", "dL_drotations"):
        ref = o64[k].astype(np.float64); got = h[k].astype(np.float64).reshape(ref.shape)
        print(f"  {k:14s} max|ref| {np.abs(ref).max():.3e}  hip err {np.abs(got-ref).max():.3e}  fp32-oracle err {np.abs(o32[k].astype(np.float64)-ref).max():.3e}")
