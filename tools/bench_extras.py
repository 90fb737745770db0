#!/usr/bin/env python
"""tools/bench_extras.py -- the side legs rounds 1-5's bench.py printed inside its one JSON line (which grew to 20.7 KB and stopped
parsing): `python bench.py --extras [PATH]` runs them after the headline and writes them to PATH
(default gpurun_out/bench_report.json); none of them is part of the contract's line.

  per_stage / step_algorithmic_bytes   every kernel of a step bracketed with HIP events (separate pass) next to its algorithmic bytes
  exp_mode_2                           both blend kernels with v_exp_f32
  sweep_1M_1080p                       the 1 M point with its own roofline object and stage table (rounds 1-2's headline)
  baseline_configs                     BASELINE.json's cfg2 / cfg3 shapes, pipelined protocol
  shell_scene_1080p                    a second occlusion regime (scenes.synth_shell: R_eff ~ R)
  two_views_in_flight_1080p            view_parallel.distributed_step(views_in_flight=2)'s loop
  eval_fps_forward_only                the reference's only benchmark (test.py:155-168)
  training_like                        forward -> L1 + D-SSIM -> backward -> Adam, static / time-varying scenes, table on / off
  next_rows                            SURVEY 8f: fused loss, activation epilogue, per-row Adam, kNN, hex-plane field, a whole iteration
"""
from __future__ import annotations

import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "saro-gs_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from bench import HBM_PEAK_GBS, Deformation, Workload, _query, roofline_of, stage_table, step_bytes, timed  # noqa: E402,F401


def exp_mode2_row(_C, wl, dev, kid):
    """Both blend kernels with exp_mode 2 (v_exp_f32, within a few ulp of mode 0: north_star's 1e-5 bar, not the bit-exact one)
    next to the default's fixed-sequence exp: event-bracketed launches of a separate, untimed pass."""
    cur = _C.get_option("exp_mode")
    res = {}
    for mode in (cur, 2):
        _C.set_option("exp_mode", mode)
        for _ in range(3):
            wl.step(None, 1)
        torch.cuda.synchronize(dev)
        _C.profile_reset()
        _C.set_option("profile", (1 << kid["blend_bwd"]) | (1 << kid["blend_fwd"]))
        t0 = time.perf_counter()
        for _ in range(10):
            wl.step(None, 1)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / 10 * 1e3
        pk = _C.profile_read()
        _C.set_option("profile", 0)
        res[f"exp_mode_{mode}"] = {"blend_fwd_ms": round(pk["blend_fwd"][0] / max(pk["blend_fwd"][1], 1), 4),
                                   "blend_bwd_ms": round(pk["blend_bwd"][0] / max(pk["blend_bwd"][1], 1), 4),
                                   "ms_per_step_with_both_bracketed": round(dt, 4)}
    _C.set_option("exp_mode", cur)
    res["note"] = "mode 2 = hardware v_exp_f32; outputs within 1e-5 of mode 0 (tests/test_gpu_parity.py::test_other_exp_modes_within_tolerance); the headline runs the mode in config.exp_mode"
    return res



def training_like_row(rast, scenes, dev, P, W, H, deg, Vs=(8, 150)):
    """The reference's call pattern (train.py:198-226, scene/saro_gaussian.py:788-829): V poses dealt round-robin, every call followed by
    loss -> backward -> Adam step (the scene changes between two visits of a pose), and -- `dynamic_opacity` -- a per-call
    trbfoutput = exp(-4 ((t - temporal_pos) / lifespan)^2) at a random timestamp t (the survival state of saro_gaussian.py:757-789:
    two calls at one camera are different scenes).  Each leg with the context's pose table on and switched off (no_order_hint = 1);
    late = Gaussians the list cut left out per call, fallbacks = forwards whose cut lists were too short and were redone."""
    import fused_adam
    import fused_loss
    _C = rast._C
    sc = scenes.synth(P, 0, sh_degree=deg)
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=dev)  # noqa: E731
    bg = t(sc["bg"])
    deform = {"dynamic_opacity": Deformation(P, dev, motion=False), "dynamic_full": Deformation(P, dev, motion=True)}
    lr = dict(xyz=1.6e-4, f_dc=2.5e-3, f_rest=1.25e-4, opacity=5e-2, scaling=5e-3, rotation=1e-3)
    m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
    gt = torch.rand(3, H, W, device=dev)
    inv = torch.ones(P, 1, device=dev)

    def settings(k, V):
        cam = scenes.camera(k, V, W, H)
        return rast.GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg, scale_modifier=1.0,
            viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=deg, campos=t(cam["campos"]), prefiltered=False)

    out = {}
    for dyn in (None, "dynamic_opacity", "dynamic_full"):
        for V in Vs:
            rasters = [rast.GaussianRasterizerRaw(settings(k, V)) for k in range(V)]
            row = {}
            for name, opt_name in (("pose_table_on", None), ("pose_table_off", "no_order_hint")):
                rc = dict(xyz=t(sc["means3D"]), rotation=t(sc["rotations"]), scaling=torch.log(t(sc["scales"])),
                          opacity=torch.logit(t(sc["opacities"]).clamp(1e-4, 1 - 1e-4)), f_dc=t(sc["shs"][:, :1]), f_rest=t(sc["shs"][:, 1:]))
                rc = {k: v.requires_grad_(True) for k, v in rc.items()}
                opt = fused_adam.GaussianAdam([{"params": [rc[k]], "lr": lr[k] * inv if k != "f_rest" else lr[k], "name": k} for k in rc], eps=1e-15)
                it = [0]

                def step():
                    mres, rres, trbf = deform[dyn].at(it[0]) if dyn else (None, None, None)
                    raster = rasters[it[0] % V]
                    it[0] += 1
                    color, _, _ = raster(rc["xyz"], m2, rc["rotation"], rc["scaling"], rc["opacity"], rc["f_dc"], rc["f_rest"],
                                         motion_residual=mres, rot_residual=rres, trbfoutput=trbf)
                    loss = fused_loss.l1_dssim_loss(color, gt, 0.2)
                    opt.zero_grad(); m2.grad = None
                    loss.backward()
                    opt.step()

                if opt_name:
                    _C.set_option(opt_name, 1)
                try:
                    for _ in range(max((4 if dyn else 2) * V, 24)):         # every pose seen twice (four times where the scene varies with t: the remembered cut is a running maximum over visits)
                        step()
                    torch.cuda.synchronize(dev)
                    fb0 = _query(_C, "cut_fallbacks")
                    n, late = max(2 * V, 96), 0
                    t0 = time.perf_counter()
                    for _ in range(n):
                        step()
                        late += _query(_C, "last_late") or 0
                    torch.cuda.synchronize(dev)
                    dt = time.perf_counter() - t0
                    row[name] = {"iterations_per_s": round(n / dt, 1), "ms_per_iteration": round(dt / n * 1e3, 4), "calls": n,
                                 "late_gaussians_per_call": int(late / n), "cut_margin_x4": _query(_C, "cut_margin_x4"),
                                 "cut_fallbacks_per_100_calls": round(100.0 * ((_query(_C, "cut_fallbacks") or 0) - (fb0 or 0)) / n, 2)}
                finally:
                    if opt_name:
                        _C.set_option(opt_name, 0)
                del opt, rc
                torch.cuda.empty_cache()
            row["table_on_over_off"] = round(row["pose_table_on"]["iterations_per_s"] / row["pose_table_off"]["iterations_per_s"], 3)
            out[(dyn or "static_opacity") + f"_V{V}"] = row
    out["note"] = ("one iteration = GaussianRasterizerRaw forward -> fused L1 + D-SSIM -> backward -> GaussianAdam.step, P = %d at %dx%d; "
                   "V poses round-robin; dynamic_opacity: opacity = sigmoid(.) * exp(-4 ((t - pos) / lifespan)^2), t ~ U(0, 1) per call; "
                   "dynamic_full: the same opacity AND means / rotations / scales moved by residuals that are functions of t "
                   "(bench.py:Deformation -- what scene/saro_gaussian.py:get_deformation returns with dx = drot = dopacity = True)" % (P, W, H))
    return out


def eval_fps_row(rast, scenes, dev, P, W, H, deg):
    """The reference's only benchmark (test.py:155-168): forward-only renders under torch.no_grad(), 20 test views x 4 passes, the first 11
    views of each pass discarded, FPS = 1 / mean of the rest; each call timed wall-clock around a device synchronisation
    (renderer/__init__.py:149, :188, :202-203).  With the context's pose table on and off (the second pass on renders known poses)."""
    _C = rast._C
    sc = scenes.synth(P, 0, sh_degree=deg)
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=dev)  # noqa: E731
    ten = {k: t(sc[k]) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    m2 = torch.zeros((P, 3), device=dev)
    bg = t(sc["bg"])
    V = 20
    rasters = []
    for k in range(V):
        cam = scenes.camera(k, V, W, H)
        rasters.append(rast.GaussianRasterizer(rast.GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg, scale_modifier=1.0,
            viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=deg, campos=t(cam["campos"]), prefiltered=False)))
    out = {}
    for name, opt_name in (("pose_table_off", "no_order_hint"), ("pose_table_on", None)):
        if opt_name:
            _C.set_option(opt_name, 1)
        try:
            kept, first_pass = [], []
            with torch.no_grad():
                for p_ in range(4):
                    for k, raster in enumerate(rasters):
                        torch.cuda.synchronize(dev)
                        t0 = time.perf_counter()
                        raster(means3D=ten["means3D"], means2D=m2, opacities=ten["opacities"], shs=ten["shs"], scales=ten["scales"], rotations=ten["rotations"])
                        torch.cuda.synchronize(dev)
                        d = time.perf_counter() - t0
                        if k >= 11:
                            kept.append(d)
                            if p_ == 0:
                                first_pass.append(d)
            out[name] = {"fps": round(1.0 / float(np.mean(kept)), 1), "ms_per_view": round(float(np.mean(kept)) * 1e3, 4),
                         "first_pass_ms_per_view": round(float(np.mean(first_pass)) * 1e3, 4), "views_timed": len(kept)}
        finally:
            if opt_name:
                _C.set_option(opt_name, 0)
    # a camera PATH: 60 frames 1.5 degrees apart, every pose rendered exactly once (a test trajectory / a video: what test.py does with a
    # scene's test cameras) -- the table never holds the pose (predicted cut depths serve it); with option near_pose = 3 it borrows the previous frame's remembered ones
    VP, NF = 240, 60

    def path(first, opts):
        for k_, v_ in opts.items():
            _C.set_option(k_, v_)
        try:
            rs = []
            for k in range(first, first + NF):
                cam = scenes.camera(k, VP, W, H)
                rs.append(rast.GaussianRasterizer(rast.GaussianRasterizationSettings(
                    image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg, scale_modifier=1.0,
                    viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=deg, campos=t(cam["campos"]), prefiltered=False)))
            ds, cut = [], 0
            with torch.no_grad():
                for i, raster in enumerate(rs):
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    raster(means3D=ten["means3D"], means2D=m2, opacities=ten["opacities"], shs=ten["shs"], scales=ten["scales"], rotations=ten["rotations"])
                    torch.cuda.synchronize(dev)
                    if i >= 10:
                        ds.append(time.perf_counter() - t0)
                        cut += 1 if int(_C.context_query("last_late")) > 0 else 0
            return {"fps": round(1.0 / float(np.mean(ds)), 1), "ms_per_view": round(float(np.mean(ds)) * 1e3, 4), "frames_timed": len(ds), "frames_under_the_list_cut": cut}
        finally:
            for k_ in opts:
                _C.set_option(k_, 0)
    out["camera_path_every_pose_new"] = {"pose_table_off": path(30, {"no_order_hint": 1}), "own_slot_only": path(100, {}), "near_pose_borrowing": path(170, {"near_pose": 3}),
                                         "note": "60 frames 1.5 degrees apart on the orbit, each pose rendered once, the first 10 discarded; same timing protocol"}
    out["note"] = "forward only, torch.no_grad(), synchronised wall clock per call (test.py:155-168 protocol), P = %d at %dx%d; first_pass = views 12-20 of pass 1 (poses never seen before)" % (P, W, H)
    return out


def in_flight_row(rast, scenes, P, W, H, deg, dev, steps, warmup, lanes=2):
    """views/s with `lanes` views in flight: view k of the batch runs forward + backward on stream k % lanes.  `steps` rounds of
    `lanes` views each are timed, after `warmup` rounds; also the same views one after the other on one stream."""
    wls = [Workload(rast, scenes, P, W, H, deg, k, 8, dev) for k in range(lanes)]
    streams = [torch.cuda.Stream(dev) for _ in range(lanes)]

    def rounds(n, use_streams):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            for k, wl in enumerate(wls):
                if use_streams:
                    with torch.cuda.stream(streams[k]):
                        wl.step()
                else:
                    wl.step()
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0

    rounds(warmup, False); rounds(warmup, True)
    t_seq = rounds(steps, False)
    t_par = rounds(steps, True)
    return {"views_per_s": round(lanes * steps / t_par, 1), "views_per_s_one_stream_same_loop": round(lanes * steps / t_seq, 1),
            "lanes": lanes, "rounds": steps, "warmup_rounds": warmup,
            "note": "throughput of a batch loop with two views in flight (distributed_step(views_in_flight=2)); `value` above is one view at a time"}


def measure_point(rast, scenes, vp, P, W, H, deg, dev, steps, warmup, full=False, kind="cube", poses=1):
    """One more workload, PIPELINED protocol (K steps back to back, one synchronisation; pose table on, every pose seen before).
    full: also the roofline object and the stage table."""
    _C = rast._C
    wl = Workload(rast, scenes, P, W, H, deg, 0, max(poses, 1), dev, kind=kind, poses=poses)
    kid = {_C.lib().gsrast_profile_kernel_name(k).decode(): k for k in range(_C.lib().gsrast_profile_kernel_count())}
    for _ in range(max(warmup, 3 * poses)):
        wl.step(None, 1)
    torch.cuda.synchronize(dev)
    _C.profile_reset()
    if full:
        _C.set_option("profile", 1 << kid["blend_bwd"])
    d = timed(wl, steps, 0, None, 1, vp, dev)
    out = {"views_per_s": round(steps / d, 3), "ms_per_step": round(d / steps * 1e3, 4), "steps": steps, "warmup": warmup, "poses": poses}
    if full:
        prof = _C.profile_read()
        _C.set_option("profile", 0)
        st = wl.stats()
        bwd_ms = prof["blend_bwd"][0] / max(prof["blend_bwd"][1], 1)
        out["config"] = {"gaussians": P, "width": W, "height": H, "instances_R": st["R"], "instances_listed": st["R_listed"],
                         "column_runs_Q": st["Q"], "R_eff": st["R_eff"], "R_eff_listed": st["R_eff_listed"], "visible": st["P_vis"]}
        out["roofline"] = roofline_of(st, bwd_ms, P)
        out["per_stage"], _ = stage_table(_C, wl, st, P, deg, H)
    del wl
    torch.cuda.empty_cache()
    return out



def loss_row(dev, H, W):
    """"Next" row: fused L1 + D-SSIM loss fwd+bwd at the bench resolution, next to the
    reference's own formulation (five depthwise conv2d + autograd) run through PyTorch on the same GPU."""
    import math
    import torch.nn.functional as F
    import fused_loss
    torch.manual_seed(0)
    y = torch.rand(3, H, W, device=dev)
    x = (y + 0.05 * torch.randn(3, H, W, device=dev)).clamp(0, 1).requires_grad_(True)

    def fused():
        x.grad = None
        fused_loss.l1_dssim_loss(x, y, 0.2).backward()

    g = torch.tensor([math.exp(-(i - 5) ** 2 / float(2 * 1.5 ** 2)) for i in range(11)], device=dev)
    g = g / g.sum()
    w = (g[:, None] @ g[None, :]).expand(3, 1, 11, 11).contiguous()

    def eager():
        x.grad = None
        conv = lambda a: F.conv2d(a[None], w, padding=5, groups=3)[0]  # noqa: E731
        mu1, mu2 = conv(x), conv(y)
        s1, s2, s12 = conv(x * x) - mu1 * mu1, conv(y * y) - mu2 * mu2, conv(x * y) - mu1 * mu2
        sm = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
        (0.8 * (x - y).abs().mean() + 0.2 * (1 - sm.mean())).backward()

    def t(fn, n=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / n * 1e3

    ms_f, ms_e = t(fused), t(eager)
    nbytes = 3 * H * W * 4 * (2 + 3 + 3 + 2 + 1)       # fwd: 2 images in, 3 maps out; bwd: 3 maps + 2 images in, grad out
    return {"fused_l1_dssim_fwd_bwd": {"ms": round(ms_f, 4), "algorithmic_MB": round(nbytes / 1e6, 1),
                                       "GBps": round(nbytes / (ms_f * 1e-3) / 1e9, 1),
                                       "hbm_frac": round(nbytes / (ms_f * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                       "pytorch_conv2d_autograd_same_gpu_ms": round(ms_e, 4),
                                       "speedup_vs_pytorch": round(ms_e / ms_f, 2), "shape": [3, H, W]}}


def epilogue_row(dev, P):
    """"Next" row: fused activation / deformation epilogue fwd+bwd for P Gaussians (all
    residuals present = the dynamic stage), next to the reference's own formulation run through PyTorch."""
    import torch.nn.functional as F
    import fused_epilogue
    torch.manual_seed(0)
    M = 16
    r = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    raw = dict(xyz=r(P, 3), motion_res=0.05 * r(P, 3), rotation=r(P, 4), rot_res=0.1 * r(P, 7), scaling=r(P, 3) - 3.0,
               opacity=2.0 * r(P, 1), trbf=torch.rand(P, 1, device=dev), f_dc=r(P, 1, 3), f_rest=0.1 * r(P, M - 1, 3),
               shs_res=0.05 * r(P, M, 3))
    raw = {k: v.requires_grad_(True) for k, v in raw.items()}
    ups = [r(P, 3), r(P, 4), r(P, 3), r(P, 1), r(P, M, 3)]

    def fused():
        for v in raw.values():
            v.grad = None
        outs = fused_epilogue.activate_gaussians(raw["xyz"], raw["rotation"], raw["scaling"], raw["opacity"], raw["f_dc"], raw["f_rest"],
                                                 motion_residual=raw["motion_res"], rot_residual=raw["rot_res"],
                                                 trbfoutput=raw["trbf"], shs_residual=raw["shs_res"])
        torch.autograd.backward(outs, ups)

    def eager():      # scene/saro_gaussian.py:807-847
        for v in raw.values():
            v.grad = None
        motion = raw["xyz"] + raw["motion_res"]
        rot = F.normalize(raw["rotation"] + raw["rot_res"][:, :4])
        scale = torch.exp(raw["scaling"] + raw["rot_res"][:, 4:])
        opa = torch.sigmoid(raw["opacity"]) * raw["trbf"]
        shs = torch.cat((raw["f_dc"], raw["f_rest"]), dim=1) + raw["shs_res"]
        torch.autograd.backward((motion, rot, scale, opa, shs), ups)

    def t(fn, n=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / n * 1e3

    ms_f, ms_e = t(fused), t(eager)
    # forward: 12+12+16+28+12+4+4 in, 12+16+12+4 out (small), 12+180+192 in, 192 out (SH); backward: ~16+28+12+4+4+16+12+4 in, 16+12+28+4+4 out
    nbytes = P * (88 + 44 + 384 + 192 + 96 + 64)
    return {"ms": round(ms_f, 4), "algorithmic_MB": round(nbytes / 1e6, 1), "GBps": round(nbytes / (ms_f * 1e-3) / 1e9, 1),
            "hbm_frac": round(nbytes / (ms_f * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "pytorch_eager_same_gpu_ms": round(ms_e, 4),
            "speedup_vs_pytorch": round(ms_e / ms_f, 2), "gaussians": P, "sh_coefficients": M}


def adam_row(dev, P):
    """"Next" row (third item): one Adam step of the seven per-Gaussian groups (60 floats per
    Gaussian) with per-row learning rates in one launch, next to torch.optim.Adam(fused=True) with scalar rates
    (torch's fused Adam has no per-row rate; the reference passes a [P,1] tensor as 'lr', saro_gaussian.py:345-398)."""
    import fused_adam
    torch.manual_seed(0)
    shapes = {"xyz": (3,), "f_dc": (1, 3), "f_rest": (15, 3), "opacity": (1,), "scaling": (3,), "rotation": (4,), "temporal_pos": (1,)}
    mk = lambda: {k: torch.randn((P,) + s, device=dev).requires_grad_(True) for k, s in shapes.items()}  # noqa: E731
    pa, pb = mk(), mk()
    inv = 1.0 + torch.rand(P, 1, device=dev)
    mine = fused_adam.GaussianAdam([{"params": [pa[k]], "lr": 1e-3 * inv if k != "f_rest" else 1e-4, "name": k} for k in shapes], eps=1e-15)
    ref = torch.optim.Adam([{"params": [pb[k]], "lr": 1e-3, "name": k} for k in shapes], lr=0.0, eps=1e-15, fused=True)
    for d in (pa, pb):
        for v in d.values():
            v.grad = torch.randn_like(v)

    def t(fn, n=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / n * 1e3

    ms_f, ms_e = t(mine.step), t(ref.step)
    nbytes = P * 60 * 28 + P * 4 * 6
    return {"ms": round(ms_f, 4), "algorithmic_MB": round(nbytes / 1e6, 1), "GBps": round(nbytes / (ms_f * 1e-3) / 1e9, 1),
            "hbm_frac": round(nbytes / (ms_f * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "torch_fused_adam_scalar_lr_same_gpu_ms": round(ms_e, 4),
            "speedup_vs_pytorch": round(ms_e / ms_f, 2), "gaussians": P, "floats_per_gaussian": 60}


def iteration_row(rast, scenes, dev, P, W, H, deg):
    """A whole static-stage training iteration (train.py:190-250 with one view): raw parameters -> activation epilogue ->
    rasterizer -> L1 + D-SSIM -> backward -> Adam, (a) with this repository's fused pieces around the rasterizer,
    (b) with the reference's PyTorch formulation of those pieces around the SAME rasterizer."""
    import math
    import torch.nn.functional as F
    import fused_adam
    import fused_epilogue
    import fused_loss
    sc = scenes.synth(P, 0, sh_degree=deg)
    cam = scenes.camera(0, 1, W, H)
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=dev)  # noqa: E731
    rs = rast.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=t(sc["bg"]), scale_modifier=1.0,
        viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=deg, campos=t(cam["campos"]), prefiltered=False)
    raster = rast.GaussianRasterizer(rs)

    def raw():
        d = dict(xyz=t(sc["means3D"]), rotation=t(sc["rotations"]), scaling=torch.log(t(sc["scales"])),
                 opacity=torch.logit(t(sc["opacities"]).clamp(1e-4, 1 - 1e-4)), f_dc=t(sc["shs"][:, :1]), f_rest=t(sc["shs"][:, 1:]))
        return {k: v.requires_grad_(True) for k, v in d.items()}

    gt = torch.rand(3, H, W, device=dev)
    ra, rb = raw(), raw()
    inv = torch.ones(P, 1, device=dev)
    lr = dict(xyz=1.6e-4, f_dc=2.5e-3, f_rest=1.25e-4, opacity=5e-2, scaling=5e-3, rotation=1e-3)
    opt_a = fused_adam.GaussianAdam([{"params": [ra[k]], "lr": lr[k] * inv if k != "f_rest" else lr[k], "name": k} for k in ra], eps=1e-15)
    opt_b = torch.optim.Adam([{"params": [rb[k]], "lr": lr[k], "name": k} for k in rb], lr=0.0, eps=1e-15, fused=True)
    g = torch.tensor([math.exp(-(i - 5) ** 2 / float(2 * 1.5 ** 2)) for i in range(11)], device=dev)
    g = g / g.sum()
    w = (g[:, None] @ g[None, :]).expand(3, 1, 11, 11).contiguous()
    m2 = torch.zeros((P, 3), device=dev, requires_grad=True)

    raster_raw = rast.GaussianRasterizerRaw(rs)
    rc = raw()
    opt_c = fused_adam.GaussianAdam([{"params": [rc[k]], "lr": lr[k] * inv if k != "f_rest" else lr[k], "name": k} for k in rc], eps=1e-15)

    def fused():        # the epilogue INSIDE the per-Gaussian kernels (gsrast_forward_raw / gsrast_backward_raw)
        color, _, _ = raster_raw(rc["xyz"], m2, rc["rotation"], rc["scaling"], rc["opacity"], rc["f_dc"], rc["f_rest"])
        loss = fused_loss.l1_dssim_loss(color, gt, 0.2)
        opt_c.zero_grad(); m2.grad = None
        loss.backward()
        opt_c.step()

    def two_ops():      # round 2's form: standalone fused epilogue in front of the drop-in rasterizer
        motion, rot, scale, opa, shs = fused_epilogue.activate_gaussians(ra["xyz"], ra["rotation"], ra["scaling"], ra["opacity"], ra["f_dc"], ra["f_rest"])
        color, _, _ = raster(means3D=motion, means2D=m2, opacities=opa, shs=shs, scales=scale, rotations=rot)
        loss = fused_loss.l1_dssim_loss(color, gt, 0.2)
        opt_a.zero_grad(); m2.grad = None
        loss.backward()
        opt_a.step()

    def eager():
        rot, scale, opa = F.normalize(rb["rotation"]), torch.exp(rb["scaling"]), torch.sigmoid(rb["opacity"])
        shs = torch.cat((rb["f_dc"], rb["f_rest"]), dim=1)
        x, _, _ = raster(means3D=rb["xyz"], means2D=m2, opacities=opa, shs=shs, scales=scale, rotations=rot)
        conv = lambda a_: F.conv2d(a_[None], w, padding=5, groups=3)[0]  # noqa: E731
        mu1, mu2 = conv(x), conv(gt)
        s1, s2, s12 = conv(x * x) - mu1 * mu1, conv(gt * gt) - mu2 * mu2, conv(x * gt) - mu1 * mu2
        sm = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
        loss = 0.8 * (x - gt).abs().mean() + 0.2 * (1 - sm.mean())
        opt_b.zero_grad(); m2.grad = None
        loss.backward()
        opt_b.step()

    def tm(fn, n=20):
        for _ in range(4):
            fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / n * 1e3

    ms_f, ms_2, ms_e = tm(fused), tm(two_ops), tm(eager)
    # the DYNAMIC stage's call shape: all four deformation residuals present (scene/saro_gaussian.py:807-847), rasterizer forward + backward
    # only -- the raw entry points against the standalone epilogue in front of the drop-in rasterizer
    g_ = torch.Generator(device="cpu").manual_seed(3)
    res = dict(motion_residual=(0.01 * torch.randn((P, 3), generator=g_)).to(dev).requires_grad_(True),
               rot_residual=(0.05 * torch.randn((P, 7), generator=g_)).to(dev).requires_grad_(True),
               trbfoutput=torch.rand((P, 1), generator=g_).to(dev).requires_grad_(True),
               shs_residual=(0.03 * torch.randn((P, 16, 3), generator=g_)).to(dev).requires_grad_(True))
    gcol = torch.randn((3, H, W), generator=g_).to(dev) / (3.0 * H * W)

    def clear():
        for v in list(rc.values()) + list(ra.values()) + list(res.values()) + [m2]:
            v.grad = None

    def dyn_raw():
        clear()
        color, _, _ = raster_raw(rc["xyz"], m2, rc["rotation"], rc["scaling"], rc["opacity"], rc["f_dc"], rc["f_rest"], **res)
        color.backward(gcol)

    def dyn_two_ops():
        clear()
        motion, rot, scale, opa, shs = fused_epilogue.activate_gaussians(ra["xyz"], ra["rotation"], ra["scaling"], ra["opacity"], ra["f_dc"], ra["f_rest"], **res)
        color, _, _ = raster(means3D=motion, means2D=m2, opacities=opa, shs=shs, scales=scale, rotations=rot)
        color.backward(gcol)

    ms_dr, ms_d2 = tm(dyn_raw), tm(dyn_two_ops)
    return {"ms": round(ms_f, 4), "iterations_per_s": round(1e3 / ms_f, 1),
            "dynamic_stage_call_all_residuals_fwd_bwd": {"raw_entry_points_ms": round(ms_dr, 4), "standalone_epilogue_then_rasterizer_ms": round(ms_d2, 4),
                                                         "note": "rasterizer forward + backward with motion / rotation+scale / trbf / SH residuals given; no loss, no optimizer"},
            "standalone_epilogue_then_rasterizer_ms": round(ms_2, 4), "pytorch_pieces_around_same_rasterizer_ms": round(ms_e, 4),
            "speedup": round(ms_e / ms_f, 2), "gaussians": P, "image": [H, W],
            "pieces": "GaussianRasterizerRaw (activations inside the per-Gaussian kernels) -> l1_dssim_loss -> backward -> GaussianAdam.step"}


def knn_row(dev, P):
    """"Next" row (second item): simple_knn.distCUDA2 for P points (the reference's random-init
    cube, dataset_readers.py:526), next to an exact k-d tree 3-NN on all host cores (scipy cKDTree, fp64)."""
    from scipy.spatial import cKDTree
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(0)
    pts = rng.uniform(-1.3, 1.3, size=(P, 3)).astype(np.float32)
    x = torch.from_numpy(pts).to(dev)
    for _ in range(2):
        distCUDA2(x)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        out = distCUDA2(x)
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / n * 1e3
    t0 = time.perf_counter()
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4, workers=-1)
    cpu_s = time.perf_counter() - t0
    want = (np.sort(d ** 2, axis=1)[:, 1:]).sum(1) / 3.0
    err = float(np.abs(out.cpu().numpy().astype(np.float64) - want).max() / want.max())
    return {"ms": round(ms, 3), "points": P, "scipy_ckdtree_all_cores_s": round(cpu_s, 3), "cores": os.cpu_count(),
            "max_abs_err_rel_to_max": err}


def hexplane_row(dev, P):
    """"Next" row (first item): the residual field's mip-mapped plane lookup for P points, forward and
    backward to the planes, at the two shipped field shapes (configs/dnerf/*.json: 64^3 x 128 frames; configs/neural_3D/*.json:
    512^3 x 256 frames; 32 features, one scale), next to the same computation spelled with PyTorch ops on the same GPU
    (avg_pool2d pyramid + grid_sample per level + lerp -- the structure tests/test_oracle_texture.py pins the oracle with)."""
    import itertools
    import torch.nn.functional as F
    import fused_hexplane
    coo = list(itertools.combinations(range(4), 2))
    g = torch.Generator(device="cpu").manual_seed(0)
    out = {}
    for tag, reso in (("dnerf_64x64x64x128", [64, 64, 64, 128]), ("neural3d_512x512x512x256", [512, 512, 512, 256])):
        C = 32
        grids = [torch.randn((1, C, reso[b], reso[a]), generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
                 for (a, b) in coo]
        pts = torch.rand((P, 4), generator=g).to(dev)
        levels = torch.cat([torch.rand((P, 3), generator=g) * float(np.log2(reso[0])), torch.zeros((P, 1))], dim=1).to(dev)
        dy = torch.randn((P, C), generator=g).to(dev)

        def ours():
            o = fused_hexplane.interpolate_ms_features(pts, [grids], 2, True, levels, None)
            o.backward(dy)
            return o

        def torch_ops():
            acc = 0
            for ci, (a, b) in enumerate(coo):
                mm = 7 if b != 3 else 0
                mips = [grids[ci]]
                while mips[-1].shape[2] > 1 and len(mips) - 1 < mm:
                    mips.append(F.avg_pool2d(mips[-1], 2))
                n = len(mips) - 1
                fl = torch.minimum(levels[:, a], levels[:, b]).clamp(0.0, float(n))
                l0 = fl.floor().long()
                l1 = torch.clamp(l0 + 1, max=n)
                f = (fl - l0)[:, None]
                grid = (2.0 * pts[:, [a, b]] - 1.0)[None, None]
                va = torch.zeros((P, C), device=dev)
                vb = torch.zeros((P, C), device=dev)
                for l in range(n + 1):          # every level sampled for the points that use it
                    ma, mb = l0 == l, l1 == l
                    if bool(ma.any()) or bool(mb.any()):
                        sm = F.grid_sample(mips[l], grid, mode="bilinear", padding_mode="border", align_corners=False)[0, :, 0].t()
                        va = torch.where(ma[:, None], sm, va)
                        vb = torch.where(mb[:, None], sm, vb)
                acc = acc + va + f * (vb - va)
            acc.backward(dy)
            return acc

        res = {}
        for name, fn, reps in (("ms", ours, 10), ("torch_ops_same_gpu_ms", torch_ops, 2)):
            for _ in range(2 if fn is ours else 1):
                for gr in grids: gr.grad = None
                o = fn()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(reps):
                for gr in grids: gr.grad = None
                o = fn()
            torch.cuda.synchronize(dev)
            res[name] = round((time.perf_counter() - t0) / reps * 1e3, 3)
            res["_out_" + name] = o.detach()
            res["_grad_" + name] = grids[0].grad.detach().clone()
        a_, b_ = res.pop("_out_ms"), res.pop("_out_torch_ops_same_gpu_ms")
        ga, gb = res.pop("_grad_ms"), res.pop("_grad_torch_ops_same_gpu_ms")
        res["max_abs_diff_vs_torch_ops"] = float((a_ - b_).abs().max())
        res["plane_grad_rel_diff_vs_torch_ops"] = float((ga - gb).abs().max() / gb.abs().max())
        # forward alone, and the gather it performs: 8 texels x 128 B per point and plane
        with torch.no_grad():
            for _ in range(2):
                fused_hexplane.interpolate_ms_features(pts, [grids], 2, True, levels, None)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(10):
                fused_hexplane.interpolate_ms_features(pts, [grids], 2, True, levels, None)
            torch.cuda.synchronize(dev)
            res["forward_ms"] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
        res["points"] = P
        out[tag] = res
        del grids
    return out



def run(a, rast, scenes, vp, dev, core: dict) -> dict:
    """Every side leg; `core` = bench.py's result (plus per_stage / stats of the headline workload)."""
    _C = rast._C
    P, W, H, deg, n_poses = a.gaussians, a.width, a.height, a.sh_degree, max(a.poses, 1)
    kid = {_C.lib().gsrast_profile_kernel_name(k).decode(): k for k in range(_C.lib().gsrast_profile_kernel_count())}
    out = {"headline": {k: core.get(k) for k in ("metric", "value", "value_warm", "ms_per_step", "ms_per_step_warm", "pipelined", "config", "roofline", "roofline_fwd")},
           "per_stage": core.get("per_stage")}
    if core.get("per_stage") and core.get("stats"):
        out["step_algorithmic_bytes"] = step_bytes(core["stats"], core["per_stage"], P, deg, core["ms_per_step"])

    def leg(name, fn):
        try:
            out[name] = fn()
        except Exception as e:      # noqa: BLE001
            out[name] = {"error": str(e)}
        torch.cuda.empty_cache()

    def exp2():
        wl = Workload(rast, scenes, P, W, H, deg, 0, n_poses, dev, kind=a.scene, poses=n_poses)
        try:
            return exp_mode2_row(_C, wl, dev, kid)
        finally:
            del wl
    leg("exp_mode_2", exp2)
    leg("sweep_1M_1080p", lambda: measure_point(rast, scenes, vp, 1_000_000, W, H, deg, dev, a.steps, a.warmup, full=True, poses=n_poses))
    leg("baseline_configs", lambda: {tag: measure_point(rast, scenes, vp, p, w, h, deg, dev, a.steps, a.warmup, poses=n_poses)
                                     for tag, p, w, h in (("cfg2_100k_800x800", 100_000, 800, 800), ("cfg3_1M_1352x1014", 1_000_000, 1352, 1014))})
    leg("shell_scene_1080p", lambda: measure_point(rast, scenes, vp, 1_000_000, W, H, deg, dev, a.steps, a.warmup, full=True, kind="shell", poses=n_poses))
    leg("two_views_in_flight_1080p", lambda: in_flight_row(rast, scenes, P, W, H, deg, dev, a.steps, a.warmup))
    leg("eval_fps_forward_only", lambda: eval_fps_row(rast, scenes, dev, P, W, H, deg))
    leg("training_like", lambda: training_like_row(rast, scenes, dev, P, W, H, deg))
    Pn = min(P, 1_000_000)      # the SURVEY 8f rows are quoted at 1 M Gaussians
    nxt = {}
    for name, fn in (("fused_l1_dssim_fwd_bwd", lambda: loss_row(dev, H, W).get("fused_l1_dssim_fwd_bwd")),
                     ("fused_activation_epilogue_fwd_bwd", lambda: epilogue_row(dev, Pn)), ("per_row_lr_adam_step", lambda: adam_row(dev, Pn)),
                     ("knn3_mean_dist2", lambda: knn_row(dev, Pn)), ("hexplane_field_fwd_bwd", lambda: hexplane_row(dev, Pn)),
                     ("static_stage_training_iteration", lambda: iteration_row(rast, scenes, dev, Pn, W, H, deg))):
        try:
            nxt[name] = fn()
        except Exception as e:      # noqa: BLE001
            nxt[name] = {"error": str(e)}
        torch.cuda.empty_cache()
    out["next_rows"] = nxt
    return out
