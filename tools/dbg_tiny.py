import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "saro-gs_amd")); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import diff_gaussian_rasterization_ch3 as rast
import scenes
from gpu_harness import run_hip
rast._C.set_option("debug_sync", 1)
dev = torch.device("cuda:0")
sc = scenes.synth(64, 3); cam = scenes.camera(0, 1, 64, 48)
g = scenes.upstream_grad(48, 64, 7)
for clip in (0, 1, 1):
    h = run_hip(rast, sc, cam, dev, dL_dcolor=g, tile_clip=clip)
    print("ok", clip, h["R"], flush=True)
sc = scenes.synth(5000, 3); cam = scenes.camera(0, 1, 320, 240)
g = scenes.upstream_grad(240, 320, 7)
h = run_hip(rast, sc, cam, dev, dL_dcolor=g, tile_clip=1); print("ok big", h["R"], flush=True)
