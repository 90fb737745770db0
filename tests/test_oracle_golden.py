"""CPU: pins the oracle (and the scene/camera builders) against golden vectors.

ref_python_vectors.npz was produced by importing the REFERENCE's own Python
(utils/sh_utils.py:eval_sh, utils/graphics_utils.py, scene/cameras.py:90-101 composition) --
see tests/golden/make_golden.py.  oracle_scene_*.npz are the oracle's own committed outputs
(drift guard: a change of the oracle's arithmetic must be deliberate)."""
import glob
import os

import numpy as np

G = os.path.join(os.path.dirname(__file__), "golden")


def test_sh_colour_matches_reference_eval_sh(orc):
    z = np.load(os.path.join(G, "ref_python_vectors.npz"))
    for deg in range(4):
        want = z[f"sh_deg{deg}_rgb_plus_half"]                       # float64, reference Python
        got64 = orc.sh_to_rgb(deg, z["sh_pos"], z["sh_campos"], z["sh_coeffs"], f64=True)
        got32 = orc.sh_to_rgb(deg, z["sh_pos"], z["sh_campos"], z["sh_coeffs"], f64=False)
        # the kernel's SH constants are fp32 literals (auxiliary.h:22-39): 1e-7 relative vs Python doubles
        np.testing.assert_allclose(got64, want, rtol=0, atol=2e-6)
        np.testing.assert_allclose(got32, want, rtol=0, atol=2e-5)
    lo, hi = z["rgb2sh_of_0_and_1"]
    assert abs(lo + 1.7725) < 1e-3 and abs(hi - 1.7725) < 1e-3       # synth()'s DC range = RGB2SH([0,1])


def test_camera_builders_match_reference_graphics_utils(scenes):
    z = np.load(os.path.join(G, "ref_python_vectors.npz"))
    for k in range(len(z["cam_R"])):
        R, T = z["cam_R"][k], z["cam_T"][k]
        fovx, fovy = z["cam_fov"][k]
        view = scenes.world_to_view(R, T)
        np.testing.assert_allclose(view.T, z["cam_world_view"][k], atol=1e-6)
        proj = scenes.projection(0.01, 100.0, float(fovx), float(fovy))
        np.testing.assert_allclose(proj.T, z["cam_projection"][k], atol=1e-6)
        full = view.T.astype(np.float32) @ proj.T.astype(np.float32)
        np.testing.assert_allclose(full, z["cam_full_proj"][k], rtol=1e-5, atol=1e-5)
        center = np.linalg.inv(view.T.astype(np.float64))[3, :3]
        np.testing.assert_allclose(center, z["cam_center"][k], atol=1e-4)
    # the reference's projection uses (zf+zn)/(zf-zn), not upstream 3DGS's zf/(zf-zn) (graphics_utils.py:70-71)
    assert abs(z["cam_projection"][0][2, 2] - (100.0 + 0.01) / (100.0 - 0.01)) < 1e-6


def test_point_projection_matches_reference_geom_transform_points(orc, scenes):
    """The oracle's screen positions and depths (forward.cu:193-198, :216, auxiliary.h ndc2Pix) against the reference's own
    Python projection `graphics_utils.geom_transform_points` of the same points through the same matrices."""
    z = np.load(os.path.join(G, "ref_python_vectors.npz"))
    pts = z["proj_points"]
    P, W, H = len(pts), 200, 144
    sc = scenes.synth(P, 5, sh_degree=0)
    sc["means3D"] = pts.astype(np.float32)
    sc["bg"] = np.zeros(3, np.float32)
    seen = 0
    for k in range(len(z["cam_R"])):
        fovx, fovy = z["cam_fov"][k]
        cam = dict(image_height=H, image_width=W, tanfovx=float(np.tan(fovx / 2)), tanfovy=float(np.tan(fovy / 2)), scale_modifier=1.0,
                   viewmatrix=z["cam_world_view"][k].astype(np.float32), projmatrix=z["cam_full_proj"][k].astype(np.float32),
                   campos=z["cam_center"][k].astype(np.float32), prefiltered=False)
        st = orc.forward(sc, cam)                     # fp32 build, as the kernel computes
        vis = st["radii"] > 0
        seen += int(vis.sum())
        ndc, view = z["proj_ndc"][k], z["proj_view"][k]
        want = np.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], axis=1)
        np.testing.assert_allclose(st["means2D"][vis], want[vis], rtol=0, atol=2e-3)      # fp32 matrices, pixels
        np.testing.assert_allclose(st["depths"][vis], view[vis, 2], rtol=2e-6, atol=2e-5)
    assert seen > 100


def _load_scene(f):
    z = np.load(f)
    sc = {k[3:]: z[k] for k in z.files if k.startswith("sc_")}
    sc["sh_degree"] = int(z["sh_degree"])
    cam = {k[4:]: z[k] for k in z.files if k.startswith("cam_")}
    for k in ("image_height", "image_width"):
        cam[k] = int(cam[k])
    for k in ("tanfovx", "tanfovy", "scale_modifier"):
        cam[k] = float(cam[k])
    return z, sc, cam


def test_oracle_reproduces_its_committed_outputs(orc):
    files = sorted(glob.glob(os.path.join(G, "oracle_scene_*.npz")))
    assert len(files) >= 2
    for f in files:
        z, sc, cam = _load_scene(f)
        o32 = orc.render(sc, cam, z["dL_dcolor"])
        for k in ("radii", "tiles_touched", "point_list", "ranges", "keys_sorted", "n_contrib"):
            np.testing.assert_array_equal(o32[k], z[k], err_msg=k)
        for k in ("out_color", "out_depth", "final_T"):
            np.testing.assert_array_equal(o32[k].view(np.uint32), z[k].view(np.uint32), err_msg=k)
        o64 = orc.render(sc, cam, z["dL_dcolor"], f64=True)
        for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"):
            np.testing.assert_allclose(o64[k], z["f64_" + k], rtol=1e-9, atol=1e-12, err_msg=k)


def _cov_scene(scenes, z, W=160, H=128):
    """The golden Gaussians placed in front of a camera (all visible), so the oracle / the kernel evaluate computeCov3D on them."""
    scales, rots = z["cov_scales"], z["cov_rotations"]
    n = len(scales)
    sc = scenes.synth(n, 77, sh_degree=0)
    sc["scales"], sc["rotations"] = scales, rots
    rng = np.random.default_rng(5)
    sc["means3D"] = rng.uniform(-0.4, 0.4, size=(n, 3)).astype(np.float32)
    return sc, scenes.camera(0, 4, W, H)


def test_cov3d_matches_reference_build_covariance(orc, scenes):
    """cov3D (6 floats: xx, xy, xz, yy, yz, zz) against the reference's own Python: strip_symmetric(L @ L^T) with
    L = build_scaling_rotation(modifier * scaling, rotation) (utils/general_utils.py:113-205 as composed in
    scene/saro_gaussian.py:33-37) -- pins the packing, the quaternion convention and Sigma = R S S^T R^T of computeCov3D
    (forward.cu:118-152) in the oracle."""
    z = np.load(os.path.join(G, "ref_python_vectors.npz"))
    assert len(z["cov_scales"]) >= 256
    sc, cam = _cov_scene(scenes, z)
    for tag, mod in (("1", 1.0), ("0p7", 0.7)):
        cam["scale_modifier"] = mod
        want = z["cov3D_mod" + tag].astype(np.float64)               # reference, fp32 arithmetic in torch
        o32 = orc.forward(sc, cam)
        o64 = orc.forward(sc, cam, st32=o32)
        vis = o32["radii"] > 0
        assert vis.mean() > 0.9
        tol = 2e-6 * np.abs(want).max(axis=1, keepdims=True)          # two fp32 evaluations in different operation orders
        assert (np.abs(o32["cov3D"] - want)[vis] <= tol[vis]).all()
        assert (np.abs(o64["cov3D"] - want)[vis] <= tol[vis]).all()
    # the rotation matrix itself: R[i][j] of build_rotation vs the oracle's cov3D for unit scales would lose the sign
    # information, so check the axis images directly: Sigma = R diag(s^2) R^T with s = (2, 1, 0.5) identifies R up to sign per column
    R = z["cov3D_mod1_R"].astype(np.float64)
    s = np.array([2.0, 1.0, 0.5])
    sc2, cam2 = _cov_scene(scenes, z)
    sc2["scales"] = np.tile(s.astype(np.float32), (len(R), 1))
    o32 = orc.forward(sc2, cam2)
    S = np.einsum("nij,j,nkj->nik", R, s * s, R)
    want6 = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], axis=1)
    vis = o32["radii"] > 0
    np.testing.assert_allclose(o32["cov3D"][vis], want6[vis], rtol=0, atol=4e-6)


def test_loss_oracle_matches_reference_loss_utils():
    """oracle/loss_oracle.py against the reference's own l1_loss / ssim (utils/loss_utils.py:18-68) combined as
    helper_train.py:50-53, evaluated by importing the reference's Python (tests/golden/make_golden.py)."""
    from oracle import loss_oracle
    z = np.load(os.path.join(G, "loss_vectors.npz"))
    np.testing.assert_array_equal(loss_oracle.window2d().astype(np.float32).view(np.uint32), z["ref_window_2d"].view(np.uint32))
    for tag in ("a", "b", "c"):
        img, gt, lam = z[f"ref_{tag}_img"], z[f"ref_{tag}_gt"], float(z[f"ref_{tag}_lambda"])
        got = loss_oracle.l1_dssim(img, gt, lam)
        np.testing.assert_allclose(got, z[f"ref_{tag}_loss_l1_ssim"], rtol=0, atol=2e-6)      # the reference ran in fp32
        assert abs(got[0] - float(z[f"ref_{tag}_loss_f64"])) < 1e-9                           # ... and in fp64


def test_focal_from_tan_fov_matches_reference_fov2focal(orc, scenes):
    """rasterizer_impl.cu:222-223: focal = size / (2 tan_fov), with tan_fov = tan(0.5 FoV) (renderer/__init__.py:50-51) -- the
    reference's own utils/graphics_utils.py fov2focal / focal2fov (tests/golden/ref_focal_vectors.npz).  Checked on the quantity the
    oracle derives from it: a Gaussian's screen-space covariance, cov2D = (focal s / z)^2 + 0.3 on the optical axis."""
    import math
    import os
    v = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_focal_vectors.npz"))
    np.testing.assert_allclose(v["pixels"] / (2.0 * np.tan(0.5 * v["fov"])), v["ref_fov2focal"], rtol=1e-14)
    np.testing.assert_allclose(v["ref_focal2fov_of_that"], v["fov"], rtol=1e-13)
    for k in range(0, 64, 9):
        fov, size = float(v["fov"][k]), int(v["pixels"][k]) | 1            # odd size: ndc 0 is a pixel centre
        size = min(size, 257)
        focal = size / (2.0 * math.tan(0.5 * fov))
        np.testing.assert_allclose(focal, float(v["ref_fov2focal"][k]) * size / float(v["pixels"][k]), rtol=1e-12)
        view = np.eye(4, dtype=np.float32)
        proj = scenes.projection(0.01, 100.0, fov, fov)
        cam = dict(image_height=size, image_width=size, tanfovx=math.tan(0.5 * fov), tanfovy=math.tan(0.5 * fov), viewmatrix=view,
                   projmatrix=np.ascontiguousarray(view @ proj.T.astype(np.float32)), campos=np.zeros(3, np.float32), scale_modifier=1.0,
                   prefiltered=False)
        s, z = 0.05, 3.0
        sc = dict(means3D=np.array([[0, 0, z]], np.float32), scales=np.full((1, 3), s, np.float32), rotations=np.array([[1, 0, 0, 0]], np.float32),
                  opacities=np.full((1, 1), 0.5, np.float32), shs=np.zeros((1, 16, 3), np.float32), sh_degree=0, bg=np.zeros(3, np.float32))
        o = orc.render(sc, cam, f64=True)
        cov = (focal * s / z) ** 2 + 0.3
        np.testing.assert_allclose(o["conic_opacity"][0, 0], 1.0 / cov, rtol=1e-5)
