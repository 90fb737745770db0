#!/usr/bin/env python
"""Pose table under the reference's actual call pattern (development helper).

  frozen : V poses dealt round-robin over ONE frozen scene (forward + backward), per visit `last_late` / early runs / redo / fallbacks,
           views/s with the table on, the list cut off, the whole table off.
  carry  : does a context's adaptive state survive a change of scene?  cube pose 0 -> shell scene -> cube poses 0 / 1 alternating.
  train  : V poses round-robin, GaussianRasterizerRaw -> L1 + D-SSIM -> GaussianAdam.step between calls, opacity = sigmoid(.) * trbf(t)
           with the reference's survival state exp(-4 ((t - pos) / lifespan)^2) (scene/saro_gaussian.py:757-789) at a random timestamp
           per call; views/s, late Gaussians, fallbacks per 100 calls; with the table on / cut off / table off.

usage: tools/pose_cycle_probe.py [frozen|carry|train|all] [P] [V,V,...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "saro-gs_amd")]
import numpy as np
import torch
import scenes
import diff_gaussian_rasterization_ch3 as rast

what = sys.argv[1] if len(sys.argv) > 1 else "all"
P = int(float(sys.argv[2])) if len(sys.argv) > 2 else 3_000_000
Vs = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 8, 32]
W, H, deg = 1920, 1080, 3
dev = torch.device("cuda:0")
_C = rast._C
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731


def settings(k, V, bg):
    cam = scenes.camera(k, V, W, H)
    return rast.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg, scale_modifier=1.0,
        viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=deg, campos=t(cam["campos"]), prefiltered=False)


MODES = (("table_on", {}), ("no_list_cut", {"no_list_cut": 1}), ("table_off", {"no_order_hint": 1}))


def q(name):
    return int(_C.context_query(name))


class Frozen:
    def __init__(self, P, kind="cube"):
        sc = scenes.synth(P, 0, sh_degree=deg) if kind == "cube" else scenes.synth_shell(P, 0, sh_degree=deg)
        self.P = P
        self.bg = t(sc["bg"])
        self.leaves = {k: t(sc[k]).requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
        self.m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
        self.g = t(scenes.upstream_grad(H, W, 1))

    def step(self, rs):
        L = self.leaves
        for p in list(L.values()) + [self.m2]:
            p.grad = None
        color, radii, depth = rast.GaussianRasterizer(rs)(means3D=L["means3D"], means2D=self.m2, opacities=L["opacities"], shs=L["shs"],
                                                          scales=L["scales"], rotations=L["rotations"])
        color.backward(self.g)


def run_cycle(step, poses, n_warm_cycles, n_timed_cycles, trace=False):
    V = len(poses)
    lates, fbs = [], []
    for c in range(n_warm_cycles):
        for k in range(V):
            step(poses[k])
            if trace:
                lates.append(q("last_late"))
    torch.cuda.synchronize(dev)
    fb0, redo0 = q("cut_fallbacks"), q("redo_count")
    n = 0
    late_sum = 0
    t0 = time.perf_counter()
    for c in range(n_timed_cycles):
        for k in range(V):
            step(poses[k])
            late_sum += q("last_late")          # host-side atomics only: no device wait
            n += 1
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    return dict(views_per_s=round(n / dt, 1), ms=round(dt / n * 1e3, 4), calls=n, mean_late=int(late_sum / n),
                fallbacks_per_100=round(100.0 * (q("cut_fallbacks") - fb0) / n, 2), redo=q("redo_count") - redo0,
                early_runs=q("last_early_runs"), runs=q("last_runs"), warm_trace=lates if trace else None)


def frozen():
    sc = Frozen(P)
    for V in Vs:
        poses = [settings(k, max(V, 8) if V <= 8 else V, sc.bg) for k in range(V)]     # neighbouring poses of an 8-ring (V <= 8)
        for name, opts in MODES:
            for k_, v_ in opts.items():
                _C.set_option(k_, v_)
            warm = max(2, (20 + V - 1) // V)
            timed = max(3, (60 + V - 1) // V)
            r = run_cycle(sc.step, poses, warm, timed, trace=(name == "table_on" and V <= 8))
            for k_ in opts:
                _C.set_option(k_, 0)
            print(json.dumps(dict(leg="frozen", P=P, V=V, mode=name, **r)), flush=True)


def carry():
    cube, shell = Frozen(P), Frozen(1_000_000, "shell")
    p0, p1 = settings(0, 8, cube.bg), settings(1, 8, cube.bg)
    seq = []
    for _ in range(12):
        cube.step(p0); seq.append(("cube0", q("last_late")))
    for _ in range(12):
        shell.step(p0); seq.append(("shell", q("last_late")))
    for i in range(100):
        cube.step(p0 if i % 2 == 0 else p1); seq.append(("cube%d" % (i % 2), q("last_late")))
    print(json.dumps(dict(leg="carry", P=P, late_per_call=seq)), flush=True)


def train():
    import bench
    import fused_adam
    import fused_loss
    sc = scenes.synth(P, 0, sh_degree=deg)
    bg = t(sc["bg"])
    deform = {"dynamic_opacity": bench.Deformation(P, dev, motion=False), "dynamic_full": bench.Deformation(P, dev, motion=True)}

    def raw():
        d = dict(xyz=t(sc["means3D"]), rotation=t(sc["rotations"]), scaling=torch.log(t(sc["scales"])),
                 opacity=torch.logit(t(sc["opacities"]).clamp(1e-4, 1 - 1e-4)), f_dc=t(sc["shs"][:, :1]), f_rest=t(sc["shs"][:, 1:]))
        return {k: v.requires_grad_(True) for k, v in d.items()}

    lr = dict(xyz=1.6e-4, f_dc=2.5e-3, f_rest=1.25e-4, opacity=5e-2, scaling=5e-3, rotation=1e-3)
    m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
    gt = torch.rand(3, H, W, device=dev)
    legs = [x for x in os.environ.get("LEGS", "static,dynamic_opacity,dynamic_full").split(",") if x]
    for dyn in legs:
        for V in Vs:
            poses = [settings(k, max(V, 8), bg) for k in range(V)]
            for name, opts in MODES:
                rc = raw()
                inv = torch.ones(P, 1, device=dev)
                opt = fused_adam.GaussianAdam([{"params": [rc[k]], "lr": lr[k] * inv if k != "f_rest" else lr[k], "name": k} for k in rc], eps=1e-15)
                for k_, v_ in opts.items():
                    _C.set_option(k_, v_)
                it = [0]

                def step(rs):
                    mres, rres, trbf = deform[dyn].at(it[0]) if dyn != "static" else (None, None, None)
                    it[0] += 1
                    color, _, _ = rast.GaussianRasterizerRaw(rs)(rc["xyz"], m2, rc["rotation"], rc["scaling"], rc["opacity"], rc["f_dc"], rc["f_rest"],
                                                                 motion_residual=mres, rot_residual=rres, trbfoutput=trbf)
                    loss = fused_loss.l1_dssim_loss(color, gt, 0.2)
                    opt.zero_grad(); m2.grad = None
                    loss.backward()
                    opt.step()

                warm = max(4 if dyn != "static" else 2, (20 + V - 1) // V)
                timed = max(2, (100 + V - 1) // V)
                r = run_cycle(step, poses, warm, timed)
                r["cut_margin_x4"] = q("cut_margin_x4"); r["cut_pause"] = q("cut_pause")
                for k_ in opts:
                    _C.set_option(k_, 0)
                print(json.dumps(dict(leg="train", scene=dyn, P=P, V=V, mode=name, **r)), flush=True)
                del opt, rc
                torch.cuda.empty_cache()


if what in ("carry", "all"):
    carry()
if what in ("frozen", "all"):
    frozen()
if what in ("train", "all"):
    train()
