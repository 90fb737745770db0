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
