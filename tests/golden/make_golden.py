"""Generates the committed golden vectors.  Run in the BUILD container (needs /root/reference):

    python tests/golden/make_golden.py

1. ref_python_vectors.npz -- outputs of the REFERENCE's own Python for the pieces of the hot path
   that exist in Python: utils/sh_utils.py:eval_sh (the SH polynomial the CUDA kernel
   forward.cu:20-71 implements) and utils/graphics_utils.py getWorld2View2 / getProjectionMatrix
   plus the scene/cameras.py:90-101 composition (the camera conventions every rasterizer input
   obeys) and graphics_utils.geom_transform_points (the point projection of forward.cu:193-198).  These pin oracle/gsrast_oracle.c:sh_to_rgb and saro-gs_amd/scenes.py against the
   reference itself.  Only inputs and outputs are stored -- no reference source.
2. oracle_scene_*.npz -- small seeded scenes with the ORACLE's outputs (fp32 build for the
   bit-exact quantities, fp64 build for the gradients).  They guard against oracle drift and give
   the GPU tests a fixture that does not depend on the oracle library being rebuilt.
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "saro-gs_amd"))
REF = "/root/reference"


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_python_vectors():
    import torch
    sh_utils = _load("ref_sh_utils", os.path.join(REF, "utils", "sh_utils.py"))
    gfx = _load("ref_graphics_utils", os.path.join(REF, "utils", "graphics_utils.py"))
    rng = np.random.default_rng(1234)
    out = {}
    # --- eval_sh: [n, 3, 16] coefficients (channel-major there), unit dirs ---
    n = 257
    pos = rng.uniform(-2, 2, size=(n, 3)).astype(np.float32)
    campos = np.array([0.3, -1.1, 3.7], np.float32)
    sh = rng.normal(0, 0.5, size=(n, 16, 3)).astype(np.float32)      # rasterizer layout [n, M, 3]
    d = pos.astype(np.float64) - campos.astype(np.float64)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    out["sh_pos"], out["sh_campos"], out["sh_coeffs"] = pos, campos, sh
    for deg in range(4):
        res = sh_utils.eval_sh(deg, torch.from_numpy(sh.astype(np.float64)).permute(0, 2, 1), torch.from_numpy(d))
        out[f"sh_deg{deg}_rgb_plus_half"] = (res + 0.5).numpy()    # forward.cu:60 adds 0.5 before clamping
    out["rgb2sh_of_0_and_1"] = np.array([float(sh_utils.RGB2SH(0.0)), float(sh_utils.RGB2SH(1.0))])
    # --- camera conventions ---
    Rs, Ts, fovs, views, projs, fulls, centers = [], [], [], [], [], [], []
    for k in range(6):
        A = rng.normal(size=(3, 3))
        Q, _ = np.linalg.qr(A)
        if np.linalg.det(Q) < 0:
            Q[:, 0] = -Q[:, 0]
        T = rng.uniform(-3, 3, size=3)
        fovx, fovy = float(rng.uniform(0.4, 1.2)), float(rng.uniform(0.4, 1.2))
        wv = torch.tensor(gfx.getWorld2View2(Q, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)   # cameras.py:90
        pj = gfx.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)  # cameras.py:98
        full = (wv.unsqueeze(0).bmm(pj.unsqueeze(0))).squeeze(0)                                   # cameras.py:100
        center = wv.inverse()[3, :3]                                                               # cameras.py:101
        Rs.append(Q); Ts.append(T); fovs.append((fovx, fovy))
        views.append(wv.numpy()); projs.append(pj.numpy()); fulls.append(full.numpy()); centers.append(center.numpy())
    # --- point projection: graphics_utils.geom_transform_points (row-vector convention, + 1e-7 on w like forward.cu:195-197) ---
    pts = rng.uniform(-4, 4, size=(300, 3))
    out["proj_points"] = pts
    out["proj_ndc"] = np.array([gfx.geom_transform_points(torch.from_numpy(pts), torch.from_numpy(f).double()).numpy() for f in fulls])
    out["proj_view"] = np.array([gfx.geom_transform_points(torch.from_numpy(pts), torch.from_numpy(v).double()).numpy() for v in views])
    out.update(cam_R=np.array(Rs), cam_T=np.array(Ts), cam_fov=np.array(fovs), cam_world_view=np.array(views),
               cam_projection=np.array(projs), cam_full_proj=np.array(fulls), cam_center=np.array(centers))
    out.update(ref_cov3d_vectors())
    np.savez_compressed(os.path.join(HERE, "ref_python_vectors.npz"), **out)
    print("wrote ref_python_vectors.npz")


def ref_focal_vectors():
    """fov2focal / focal2fov of the reference's OWN utils/graphics_utils.py:76-80: pins focal = size / (2 tan(fov / 2)), the relation
    rasterizer_impl.cu:222-223 evaluates from tan_fov (focal_y = height / (2 tan_fovy)) and that scenes.camera / the oracle restate."""
    gfx = _load("ref_graphics_utils", os.path.join(REF, "utils", "graphics_utils.py"))
    rng = np.random.default_rng(77)
    fov = rng.uniform(0.2, 2.4, size=64)
    pix = rng.integers(16, 4096, size=64).astype(np.float64)
    focal = np.array([gfx.fov2focal(f, p) for f, p in zip(fov, pix)], np.float64)
    back = np.array([gfx.focal2fov(fc, p) for fc, p in zip(focal, pix)], np.float64)
    np.savez_compressed(os.path.join(HERE, "ref_focal_vectors.npz"), fov=fov, pixels=pix, ref_fov2focal=focal, ref_focal2fov_of_that=back)
    print("ref_focal_vectors.npz written")


def oracle_scenes():
    import scenes
    from oracle import oracle as orc
    orc.build()
    orc.set_exp_mode(0)
    specs = [("a", 300, 64, 48, 3, 0.8, (1, 5), 101), ("b", 1200, 112, 80, 2, 1.0, (3, 7), 202)]
    for tag, P, W, H, deg, sm, (k, V), seed in specs:
        sc = scenes.synth(P, seed, sh_degree=deg, scale_mul=sm)
        sc["bg"] = np.array([0.1, 0.2, 0.3], np.float32)
        cam = scenes.camera(k, V, W, H)
        g = scenes.upstream_grad(H, W, seed + 1) * (H * W)
        o32 = orc.render(sc, cam, g)
        o64 = orc.render(sc, cam, g, f64=True)
        d = {"sc_" + k_: v for k_, v in sc.items() if isinstance(v, np.ndarray)}
        d["sh_degree"] = np.int32(deg)
        d.update({"cam_" + k_: np.asarray(v) for k_, v in cam.items() if k_ != "prefiltered"})
        d["dL_dcolor"] = g
        for k_ in ("radii", "tiles_touched", "point_list", "ranges", "keys_sorted", "n_contrib", "out_color",
                   "out_depth", "final_T"):
            d[k_] = o32[k_]
        for k_ in ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations", "out_color"):
            d["f64_" + k_] = o64[k_]
        path = os.path.join(HERE, f"oracle_scene_{tag}.npz")
        np.savez_compressed(path, **d)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def _load_with_placeholders(name, path, absent, cuda_to_cpu=False):
    """Import one of the reference's Python files whose module-level imports name packages this image lacks.
    `absent`: module names (e.g. "cv2", "torchmetrics") that get an EMPTY placeholder module in sys.modules for the
    duration of the import -- the functions used below never touch them.  cuda_to_cpu: utils/general_utils.py
    creates its work tensors with a hard-coded device="cuda" (general_utils.py:115, :131, :190); this container has
    no GPU, so the module's `torch` name is bound to a forwarding proxy whose zeros() maps that device string to
    "cpu".  Neither placeholder supplies any arithmetic: every number stored comes out of the reference's own
    statements (the rotation-matrix entries, R @ L, L @ L^T, strip_symmetric; the window, the five conv2d, the SSIM map)."""
    import types
    saved = {}
    for m in absent:
        saved[m] = sys.modules.get(m)
        ph = types.ModuleType(m)
        if m == "torchmetrics":
            # loss_utils.py:16 imports this name and :101 instantiates it at module level (an MS-SSIM metric object that
            # l1_loss / ssim never use): an inert callable returning None
            ph.MultiScaleStructuralSimilarityIndexMeasure = lambda *a, **k: None
        if m == "PIL":
            ph.Image = None
        sys.modules[m] = ph
    try:
        mod = _load(name, path)
    finally:
        for m, old in saved.items():
            if old is None:
                sys.modules.pop(m, None)
            else:
                sys.modules[m] = old
    if cuda_to_cpu:
        import torch

        class _TorchOnCpu:
            def __getattr__(self, k):
                return getattr(torch, k)

            @staticmethod
            def zeros(*a, **kw):
                if kw.get("device") == "cuda":
                    kw["device"] = "cpu"
                return torch.zeros(*a, **kw)
        mod.torch = _TorchOnCpu()
    return mod


def ref_cov3d_vectors():
    """cov3D of the reference's OWN Python: build_covariance_from_scaling_rotation (scene/saro_gaussian.py:33-37) =
    strip_symmetric(L @ L^T), L = build_scaling_rotation(modifier * scaling, rotation) (utils/general_utils.py:113-205).
    Pins the packing order [xx, xy, xz, yy, yz, zz], the quaternion convention (r, x, y, z) and Sigma = R S S^T R^T of
    computeCov3D (forward.cu:118-152), in the oracle and in the HIP kernel.  Unit quaternions only: build_rotation
    normalises, the CUDA kernel takes the quaternion as given (forward.cu:127), and SaRO-GS always passes normalised ones."""
    import torch
    gu = _load_with_placeholders("ref_general_utils", os.path.join(REF, "utils", "general_utils.py"), ("cv2",), cuda_to_cpu=True)
    rng = np.random.default_rng(4321)
    n = 384
    scales = np.exp(rng.uniform(np.log(0.003), np.log(1.5), size=(n, 3))).astype(np.float32)
    scales[:8] = np.float32(0.25)                                  # isotropic
    scales[8:16, 0] *= np.float32(200.0)                           # needles
    q = rng.normal(size=(n, 4))
    q[16] = (1, 0, 0, 0); q[17] = (0, 1, 0, 0); q[18] = (0, 0, 1, 0); q[19] = (0, 0, 0, 1); q[20] = (-1, 0, 0, 0)
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    q = (q.astype(np.float64) / np.linalg.norm(q.astype(np.float64), axis=1, keepdims=True)).astype(np.float32)   # unit to fp32 rounding
    out = {"cov_scales": scales, "cov_rotations": q}
    for tag, mod in (("1", 1.0), ("0p7", 0.7)):
        L = gu.build_scaling_rotation(mod * torch.from_numpy(scales), torch.from_numpy(q))      # saro_gaussian.py:34
        cov = gu.strip_symmetric(L @ L.transpose(1, 2))                                          # :35-36
        out["cov3D_mod" + tag] = cov.numpy()
        # the same statements evaluated in float64 (inputs widened): tight check of the fp64 oracle build
        L64 = gu.build_scaling_rotation(mod * torch.from_numpy(scales.astype(np.float64)), torch.from_numpy(q.astype(np.float64)))
        out["cov3D_mod" + tag + "_R"] = gu.build_rotation(torch.from_numpy(q)).numpy()
    return out


def ref_loss_vectors():
    """l1_loss / ssim of the reference's OWN utils/loss_utils.py:18-68 combined as helper_train.py:50-53, plus the
    autograd gradient of that loss w.r.t. the rendered image: pins oracle/loss_oracle.py and the fused HIP loss."""
    import torch
    lu = _load_with_placeholders("ref_loss_utils", os.path.join(REF, "utils", "loss_utils.py"), ("torchmetrics",))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_loss
    out = {"ref_window_1d": lu.gaussian(11, 1.5).numpy(), "ref_window_2d": lu.create_window(11, 1)[0, 0].numpy()}     # loss_utils.py:25-35
    for tag, (C, H, W, seed, lam) in {"a": (3, 48, 64, 9, 0.2), "b": (3, 37, 53, 10, 0.2), "c": (1, 16, 16, 11, 0.5)}.items():
        img, gt = test_loss._images(C, H, W, seed)
        x = torch.from_numpy(img).clone().requires_grad_(True)
        y = torch.from_numpy(gt)
        l1 = lu.l1_loss(x, y)
        s = lu.ssim(x, y)
        loss = (1.0 - lam) * l1 + lam * (1.0 - s)                                                # helper_train.py:50-53
        loss.backward()
        out.update({f"ref_{tag}_img": img, f"ref_{tag}_gt": gt, f"ref_{tag}_lambda": np.float64(lam),
                    f"ref_{tag}_loss_l1_ssim": np.array([loss.item(), l1.item(), s.item()], np.float64),
                    f"ref_{tag}_grad": x.grad.numpy()})
        # float64 evaluation of the same statements (window stays the reference's fp32 window widened)
        x64 = torch.from_numpy(img.astype(np.float64)).requires_grad_(True)
        y64 = torch.from_numpy(gt.astype(np.float64))
        l64 = (1.0 - lam) * lu.l1_loss(x64, y64) + lam * (1.0 - lu.ssim(x64, y64))
        l64.backward()
        out[f"ref_{tag}_loss_f64"] = np.float64(l64.item())
        out[f"ref_{tag}_grad_f64"] = x64.grad.numpy()
    return out


def loss_vectors():
    """loss_vectors.npz: one image pair + the loss oracle's (loss, l1, ssim) -- drift guard for oracle/loss_oracle.py.
    (utils/loss_utils.py of the reference imports torchmetrics, which this image lacks, so it cannot be imported.)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_loss
    from oracle import loss_oracle
    img, gt = test_loss._images(3, 48, 64, 9)
    out = np.array(loss_oracle.l1_dssim(img, gt, 0.2))
    extra = {}
    path = os.path.join(HERE, "loss_vectors.npz")
    if os.path.isdir(REF):
        extra = ref_loss_vectors()
    elif os.path.exists(path):          # no reference here: keep the committed reference-derived entries
        old = np.load(path)
        extra = {k: old[k] for k in old.files if k.startswith("ref_")}
    np.savez_compressed(path, img=img, gt=gt, lambda_dssim=np.float64(0.2), loss_l1_ssim=out, **extra)
    print("wrote loss_vectors.npz", out, sorted(extra)[:4])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "focal":        # only the (round 3) focal fixture; the others stay as committed
        ref_focal_vectors()
        sys.exit(0)
    if os.path.isdir(REF):
        ref_python_vectors()
        ref_focal_vectors()
    else:
        print("no /root/reference here: keeping the committed ref_python_vectors.npz")
    oracle_scenes()
    loss_vectors()
