#!/usr/bin/env python
"""Predicted cut (option tau_cut) under the microscope (development helper): renders pose k of the 3 M cube a few times with the pose table
on / off and prints, per call, the late Gaussians, how many tiles got a cut depth and the opacity-mass table's total (read from the image buffer)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "saro-gs_amd")]
import numpy as np, torch, scenes
import diff_gaussian_rasterization_ch3 as rast
_C = rast._C
P = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000
table = int(sys.argv[2]) if len(sys.argv) > 2 else 0
tau_min = int(sys.argv[3]) if len(sys.argv) > 3 else 24
verbose = int(sys.argv[4]) if len(sys.argv) > 4 else 1
if len(sys.argv) > 5:
    _C.set_option("tau_sample", int(sys.argv[5]))
W, H = 1920, 1080
dev = torch.device("cuda:0")
sc = scenes.synth(P, 0)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
ten = {k: t(sc[k]) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
e = torch.empty(0)
gx, gy = (W + 15) // 16, (H + 15) // 16
T, N = gx * gy, W * H
al = lambda x: (x + 255) & ~255
def layout():
    o = 0; offs = {}
    Tg = gx * ((gy + 7) // 8)
    for name, nbytes in (("final_T", N * 4), ("n_contrib", N * 4), ("ranges", T * 8), ("tile_max", T * 4), ("order_fwd", T * 4), ("order_bwd", T * 4),
                         ("bucket_cnt", (9 * 64 + 80) * 4), ("bucket_list", (8 * 64 * Tg + 64 * T) * 2), ("zcut_used", T * 4), ("tile_flags", T), ("tau_hist", 4 * T * 32 * 4)):
        offs[name] = o; o = al(o + nbytes)
    return offs
L = layout()
_C.set_option("no_order_hint", 0 if table else 1)
_C.policy_event("tau_min", tau_min)
import time
cams = [scenes.camera(k, 8, W, H) for k in range(8)]
camt = [(t(c["viewmatrix"]), t(c["projmatrix"]), t(c["campos"])) for c in cams]
def fwd(k):
    c = cams[k]; vm, pm, cp = camt[k]
    return _C.rasterize_gaussians(t(sc["bg"]), ten["means3D"], e, ten["opacities"], ten["scales"], ten["rotations"], 1.0, e, vm, pm, c["tanfovx"], c["tanfovy"], H, W, ten["shs"], 3, cp, False)
if not verbose:
    for i in range(16):
        fwd(i % 8)
    torch.cuda.synchronize(); p0 = _C.context_query("completion_passes"); late = er = 0
    t0 = time.perf_counter()
    for i in range(80):
        fwd(i % 8); late += _C.context_query("last_late"); er += _C.context_query("last_early_runs")
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 80 * 1e3
    print(f"P {P} table {table} tau_min {tau_min}: forward {dt:.3f} ms, late {late // 80}, early runs {er // 80} of {_C.context_query('last_runs')}, completion passes {_C.context_query('completion_passes') - p0} in 80 calls, tau_req now {_C.context_query('tau_req')}, pause {_C.context_query('cut_pause')}")
    _C.profile_reset(); _C.set_option("profile", -1)
    for i in range(24):
        fwd(i % 8)
    torch.cuda.synchronize()
    pk = _C.profile_read(); _C.set_option("profile", 0)
    print("   stages (ms):", {k: round(v[0] / max(v[1], 1), 3) for k, v in pk.items() if v[1]})
    sys.exit(0)
_C.set_option("debug_state", 1)      # (the mass table in the call's own image buffer, not the context's self-zeroing one)
for i in range(10):
    k = i % 2
    cam = scenes.camera(k, 8, W, H)
    R, color, radii, gb, bb, ib, depth = _C.rasterize_gaussians(t(sc["bg"]), ten["means3D"], e, ten["opacities"], ten["scales"], ten["rotations"], 1.0, e,
        t(cam["viewmatrix"]), t(cam["projmatrix"]), cam["tanfovx"], cam["tanfovy"], H, W, ten["shs"], 3, t(cam["campos"]), False)
    torch.cuda.synchronize()
    z = ib[L["zcut_used"]: L["zcut_used"] + T * 4].view(torch.int32).cpu().numpy().view(np.uint32)
    th = ib[L["tau_hist"]: L["tau_hist"] + 4 * T * 32 * 4].view(torch.int32).cpu().numpy().view(np.uint32).reshape(4, T, 32).sum(0)
    print(f"call {i} pose {k}: late {_C.context_query('last_late')} early runs {_C.context_query('last_early_runs')} of {_C.context_query('last_runs')}; tiles with a cut {(z != 0xFFFFFFFF).sum()} of {T}; "
          f"mass table total {int(th.sum())}, per-bin {th.sum(0).tolist()}; tiles with mass >= 24*256: {(th.sum(1) >= 24*256).sum()}; pause {_C.context_query('cut_pause')} tau_req {_C.context_query('tau_req')}", flush=True)
