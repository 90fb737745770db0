"""-m gpu: the HIP path (through the drop-in API and the C ABI) against the CPU oracle.

Bars (BASELINE.json north_star):
  * bit-exact for every integer / index quantity: radii, tiles_touched, sorted keys
    (tile << 32 | depth bits), point_list, tile ranges, n_contrib;
  * with exp_mode 0 (the default: fixed-sequence exp shared with the oracle) the WHOLE forward is
    bit-exact -- per-Gaussian state, colour, depth, final transmittance;
  * gradients: |hip - f64 truth| <= 1e-5 abs (+1e-4 relative for the few large entries); the f64
    truth replays the fp32 control flow, so the comparison has no threshold-flip outliers.
"""
import os

import numpy as np
import pytest

from gpu_harness import bits, check_clipped_lists, run_hip

pytestmark = pytest.mark.gpu

ATOL = 1e-5   # north_star tolerance on RGB / depth / gradients
RTOL = 1e-4

CASES = [
    # name,            P,     W,   H,  deg, scale_mul, cam (k, V)
    ("tiny",           64,    64,  48, 3, 1.0, (0, 1)),
    ("ragged",         2000,  97,  83, 3, 1.0, (1, 5)),    # image not a multiple of 16
    ("deg0",           1500,  128, 96, 0, 1.0, (2, 5)),
    ("deg1",           1500,  128, 96, 1, 1.0, (3, 5)),
    ("deg2",           1500,  128, 96, 2, 0.7, (4, 5)),
    ("cfg1_10k_400",   10000, 400, 400, 3, 1.0, (0, 1)),   # BASELINE config 1
    ("dense_small",    20000, 256, 192, 3, 0.5, (1, 3)),
]


def _scene(scenes, P, seed, deg, scale_mul):
    sc = scenes.synth(P, seed, sh_degree=deg, scale_mul=scale_mul)
    return sc


def _check_forward_exact(o, h, clipped=False):
    """clipped: h was produced with tile_clip=1 -- the lists are checked as subsequences, everything else bit for bit."""
    P = o["P"]
    assert h["R"] == o["R"]          # num_rendered keeps the reference's meaning (tiles of the 3-sigma squares)
    np.testing.assert_array_equal(h["radii"], o["radii"])
    np.testing.assert_array_equal(h["tiles_touched"], o["tiles_touched"])
    if clipped:
        check_clipped_lists(o, h, o["W"], o["H"])
    else:
        np.testing.assert_array_equal(h["keys_sorted"], o["keys_sorted"])
        np.testing.assert_array_equal(h["point_list"], o["point_list"])
        np.testing.assert_array_equal(h["ranges"], o["ranges"])
    vis = o["radii"] > 0
    for k in ("depths", "means2D", "conic_opacity", "cov3D"):
        np.testing.assert_array_equal(bits(h[k][vis]), bits(o[k][vis]), err_msg=k)
    np.testing.assert_array_equal(bits(h["rgb"][vis]), bits(np.ascontiguousarray(o["colors"][vis])), err_msg="rgb")
    if "clamped" in o and o["M"] > 0:
        np.testing.assert_array_equal(h["clamped"][vis], o["clamped"][vis])
    if not clipped:
        np.testing.assert_array_equal(h["n_contrib"], o["n_contrib"])
    np.testing.assert_array_equal(bits(h["final_T"]), bits(o["final_T"]))
    np.testing.assert_array_equal(bits(h["out_color"]), bits(o["out_color"]))
    np.testing.assert_array_equal(bits(h["out_depth"]), bits(o["out_depth"]))
    assert P == len(h["radii"])


def _check_grads(o64, o32, h, names, strict=False, conditioning=False):
    """Every gradient tensor within conftest.grad_tol of the fp64 truth: max(1e-5 * max|ref| + 1e-4 * |ref|, 8 x the fp32 oracle's own
    worst error on the tensor) -- the same bar whatever the upstream gradient's scale (`strict` / `conditioning` are kept for the call
    sites' sake: rounds 1-5 used an ABSOLUTE 1e-5 with `strict`, which at the bench's N(0,1)/(3HW) upstream gradient was 40 % of the
    largest entry of dL/dsh, and applied the fp32 floor only to needle-shaped scenes)."""
    from conftest import grad_tol
    for k in names:
        ref = o64[k].astype(np.float64)
        got = h[k].astype(np.float64).reshape(ref.shape)
        err = np.abs(got - ref)
        tol = grad_tol(ref, o32[k])      # (incl. the fp32 floor: 8x the fp32 oracle's own worst error on this tensor -- see conftest.grad_tol)
        if os.environ.get("GSRAST_GRAD_REPORT"):      # development: print how far both fp32 evaluations are from the bar instead of asserting
            e32 = np.abs(o32[k].astype(np.float64) - ref)
            w = np.unravel_index(np.argmax(err / np.maximum(tol, 1e-300)), err.shape)
            print(f"GRAD_REPORT {k}: max|ref| {np.abs(ref).max():.3e}  hip worst err/tol {float((err / np.maximum(tol, 1e-300)).max()):.2f} ({int((err > tol).sum())} over; at {w}: ref {ref[w]:.3e} err {err[w]:.3e}, "
                  f"oracle32 err there {e32[w]:.3e})  oracle32 worst err/tol {float((e32 / np.maximum(tol, 1e-300)).max()):.2f} ({int((e32 > tol).sum())} over)  "
                  f"max err / max|ref|: hip {err.max() / max(np.abs(ref).max(), 1e-300):.2e} oracle32 {e32.max() / max(np.abs(ref).max(), 1e-300):.2e}", flush=True)
            continue
        assert (err <= tol).all(), f"{k}: max abs err {err.max():.3e} (max |ref| {np.abs(ref).max():.3e}), worst err / tol {float((err / np.maximum(tol, 1e-300)).max()):.2f}, {int((err > tol).sum())} entries over"
        # the fp32 oracle (different summation order) must sit in the same band
        err32 = np.abs(o32[k].astype(np.float64) - ref)
        assert (err32 <= tol).all(), f"oracle32 {k}: {err32.max():.3e}"


@pytest.mark.parametrize("name,P,W,H,deg,scale_mul,camkv", CASES, ids=[c[0] for c in CASES])
def test_forward_backward_parity(name, P, W, H, deg, scale_mul, camkv, orc, scenes, rast, gpu):
    sc = _scene(scenes, P, seed=sum(map(ord, name)) % 1000, deg=deg, scale_mul=scale_mul)
    cam = scenes.camera(camkv[0], camkv[1], W, H)
    g = scenes.upstream_grad(H, W, 7) * (H * W)       # O(1) upstream gradient: a harder test than 1/(3HW)
    orc.set_exp_mode(0)
    o32 = orc.render(sc, cam, g)
    o64 = orc.render(sc, cam, g, f64=True)
    # clip 0: the reference's literal lists; 1: the product default (row-clipped lists).  dense 1: the per-Gaussian backward reads
    # every Gaussian; 0 (default): the ones with an all-zero gradient record get their zeros without being read
    for clip, dense in ((0, 0), (1, 0), (1, 1)):
        rast._C.set_option("dense_backward", dense)
        try:
            h = run_hip(rast, sc, cam, gpu, dL_dcolor=g, exp_mode=0, tile_clip=clip)
        finally:
            rast._C.set_option("dense_backward", 0)
        _check_forward_exact(o32, h, clipped=bool(clip))
        _check_grads(o64, o32, h, ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"])


def test_bench_shaped_gradient_magnitude(orc, scenes, rast, gpu):
    """Same as above with the bench's own upstream gradient N(0,1)/(3HW)."""
    P, W, H = 5000, 320, 240
    sc = scenes.synth(P, 11)
    cam = scenes.camera(0, 1, W, H)
    g = scenes.upstream_grad(H, W, 12)
    o32 = orc.render(sc, cam, g)
    o64 = orc.render(sc, cam, g, f64=True)
    h = run_hip(rast, sc, cam, gpu, dL_dcolor=g)
    _check_forward_exact(o32, h)
    _check_grads(o64, o32, h, ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"],
                 strict=True)


def test_white_background_and_colors_precomp(orc, scenes, rast, gpu):
    """Segment-render call pattern (renderer/__init__.py:215-225): colors_precomp, shs=None; non-zero bg
    exercises the background term of dL/dalpha (backward.cu:531-534)."""
    P, W, H = 3000, 160, 120
    sc = scenes.synth(P, 21, scale_mul=0.6)
    sc["bg"] = np.array([1.0, 1.0, 1.0], np.float32)
    sc["opacities"] = (sc["opacities"] * 0.3).astype(np.float32)     # keep final_T > 0 so the bg term matters
    cam = scenes.camera(2, 7, W, H)
    rng = np.random.default_rng(5)
    cp = rng.uniform(0, 1, size=(P, 3)).astype(np.float32)
    g = scenes.upstream_grad(H, W, 22) * (H * W)
    o32 = orc.render(sc, cam, g, colors_precomp=cp)
    o64 = orc.render(sc, cam, g, f64=True, colors_precomp=cp)
    h = run_hip(rast, sc, cam, gpu, dL_dcolor=g, colors_precomp=cp)
    _check_forward_exact(o32, h)
    _check_grads(o64, o32, h, ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dcolors", "dL_dscales", "dL_drotations"])
    assert h.get("dL_dsh") is None
    # ... and the bar BITES: the same backward without the background term of dL/dalpha (option "mutate" bit 1, tests only) must be red
    rast._C.set_option("mutate", 2)
    try:
        hm = run_hip(rast, sc, cam, gpu, dL_dcolor=g, colors_precomp=cp)
    finally:
        rast._C.set_option("mutate", 0)
    _check_forward_exact(o32, hm)
    with pytest.raises(AssertionError):
        _check_grads(o64, o32, hm, ["dL_dopacity"])


def test_cov3d_precomp_path(orc, scenes, rast, gpu):
    P, W, H = 2000, 128, 128
    sc = scenes.synth(P, 31)
    cam = scenes.camera(1, 4, W, H)
    base = orc.forward(sc, cam)
    cov = np.ascontiguousarray(base["cov3D"]).astype(np.float32)
    g = scenes.upstream_grad(H, W, 32) * (H * W)
    o32 = orc.render(sc, cam, g, cov3D_precomp=cov)
    o64 = orc.render(sc, cam, g, f64=True, cov3D_precomp=cov)
    h = run_hip(rast, sc, cam, gpu, dL_dcolor=g, cov3D_precomp=cov)
    np.testing.assert_array_equal(h["radii"], o32["radii"])
    np.testing.assert_array_equal(bits(h["out_color"]), bits(o32["out_color"]))
    _check_grads(o64, o32, h, ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dcov3D"])


def test_culling_edge_cases(orc, scenes, rast, gpu):
    """Gaussians behind the camera, off-screen, huge (cover every tile) and degenerate-tiny."""
    W, H = 96, 64
    sc = scenes.synth(400, 41)
    cam = scenes.camera(0, 1, W, H)
    m = sc["means3D"]
    m[:50] *= 8.0                      # far off-screen / behind
    m[50:60] = m[50:60] * 0.01         # near the origin
    sc["scales"][60:70] *= 40.0        # huge: rect clamps to the whole grid
    sc["scales"][70:80] *= 1e-4        # tiny: the 0.3 low-pass dominates
    eye = np.linalg.inv(cam["viewmatrix"].astype(np.float64))[3, :3]
    m[80:90] = (eye + (eye / np.linalg.norm(eye)) * 2.0).astype(np.float32)  # behind the camera
    g = scenes.upstream_grad(H, W, 42) * (H * W)
    o32 = orc.render(sc, cam, g)
    o64 = orc.render(sc, cam, g, f64=True)
    h = run_hip(rast, sc, cam, gpu, dL_dcolor=g)
    assert (o32["radii"] == 0).sum() > 10 and (o32["radii"] > 0).sum() > 100
    _check_forward_exact(o32, h)
    _check_grads(o64, o32, h, ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"])
    # culled Gaussians get exactly zero gradient everywhere
    dead = o32["radii"] == 0
    for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"):
        assert not h[k][dead].any(), k


def test_all_culled_and_empty(rast, scenes, gpu):
    import torch
    from conftest import settings_from
    W, H = 64, 48
    cam = scenes.camera(0, 1, W, H)
    sc = scenes.synth(32, 51)
    sc["means3D"] = (sc["means3D"] * 0 + np.array([100.0, 0, 0], np.float32)).astype(np.float32)
    sc["bg"] = np.array([0.25, 0.5, 0.75], np.float32)
    h = run_hip(rast, sc, cam, gpu, dL_dcolor=np.ones((3, H, W), np.float32))
    assert h["R"] == 0 and not h["radii"].any()
    for c in range(3):
        assert (h["out_color"][c] == sc["bg"][c]).all()
    assert (h["out_depth"] == 15.0).all()
    assert not h["dL_dmeans3D"].any()
    # P == 0: the reference skips the rasterizer and returns zero images (rasterize_points.cu:81)
    rs = settings_from(rast, cam, sc, gpu)
    z = lambda *s: torch.zeros(*s, device=gpu)  # noqa: E731
    color, radii, depth = rast.GaussianRasterizer(rs)(means3D=z(0, 3), means2D=z(0, 3), opacities=z(0, 1),
                                                      shs=z(0, 16, 3), scales=z(0, 3), rotations=z(0, 4))
    assert color.shape == (3, H, W) and not color.any() and radii.numel() == 0 and not depth.any()


@pytest.mark.parametrize("mode", [1, 2])
def test_other_exp_modes_within_tolerance(mode, orc, scenes, rast, gpu):
    """exp_mode 1 (OCML expf) and 2 (v_exp_f32): integer binning still exact; colour within 1e-5 except
    for the rare pixels where a 1-ulp difference in exp flips a threshold (alpha < 1/255, T < 1e-4,
    median crossing) -- inherent to ANY two exp implementations, the reference's own included."""
    P, W, H = 10000, 400, 400
    sc = scenes.synth(P, 0)
    cam = scenes.camera(0, 1, W, H)
    g = scenes.upstream_grad(H, W, 1)
    orc.set_exp_mode(1)
    try:
        o32 = orc.render(sc, cam, g)
    finally:
        orc.set_exp_mode(0)
    h = run_hip(rast, sc, cam, gpu, dL_dcolor=g, exp_mode=mode)
    rast._C.set_option("exp_mode", 0)
    np.testing.assert_array_equal(h["point_list"], o32["point_list"])
    np.testing.assert_array_equal(h["ranges"], o32["ranges"])
    bad = np.abs(h["out_color"] - o32["out_color"]).max(axis=0) > ATOL
    assert bad.mean() < 1e-4, f"{bad.sum()} pixels off"
    assert (h["n_contrib"] != o32["n_contrib"]).mean() < 1e-3
    for k in ("dL_dmeans3D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"):
        err = np.abs(h[k].astype(np.float64).reshape(o32[k].shape) - o32[k])
        assert (err > ATOL + RTOL * np.abs(o32[k])).mean() < 1e-4, k


def test_mark_visible(orc, scenes, rast, gpu):
    import torch
    from conftest import settings_from
    sc = scenes.synth(5000, 61)
    sc["means3D"] *= 3.0
    cam = scenes.camera(3, 8, 128, 96)
    rs = settings_from(rast, cam, sc, gpu)
    got = rast.GaussianRasterizer(rs).markVisible(torch.as_tensor(sc["means3D"], device=gpu)).cpu().numpy()
    want = orc.mark_visible(sc["means3D"], cam["viewmatrix"], cam["projmatrix"])
    np.testing.assert_array_equal(got, want)
    assert 0 < want.sum() < len(want)


def test_golden_fixture(rast, gpu):
    """Committed inputs + expected outputs (tests/golden/oracle_scene_*.npz, made by make_golden.py)."""
    import glob, os
    from conftest import grad_tol
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "oracle_scene_*.npz")))
    assert files, "golden fixtures missing"
    for f in files:
        z = np.load(f)
        sc = {k[3:]: z[k] for k in z.files if k.startswith("sc_")}
        sc["sh_degree"] = int(z["sh_degree"])
        cam = {k[4:]: z[k] for k in z.files if k.startswith("cam_")}
        for k in ("image_height", "image_width"):
            cam[k] = int(cam[k])
        for k in ("tanfovx", "tanfovy", "scale_modifier"):
            cam[k] = float(cam[k])
        h = run_hip(rast, sc, cam, gpu, dL_dcolor=z["dL_dcolor"])
        np.testing.assert_array_equal(h["radii"], z["radii"])
        np.testing.assert_array_equal(h["point_list"], z["point_list"])
        np.testing.assert_array_equal(h["ranges"], z["ranges"])
        np.testing.assert_array_equal(h["n_contrib"], z["n_contrib"])
        np.testing.assert_array_equal(bits(h["out_color"]), bits(z["out_color"]))
        np.testing.assert_array_equal(bits(h["out_depth"]), bits(z["out_depth"]))
        for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"):
            ref = z["f64_" + k]
            err = np.abs(h[k].astype(np.float64).reshape(ref.shape) - ref)
            if os.environ.get("GSRAST_GRAD_REPORT"):
                print(f"GRAD_REPORT golden {os.path.basename(f)} {k}: worst err / tol {float((err / np.maximum(grad_tol(ref), 1e-300)).max()):.3f}", flush=True)
            assert (err <= grad_tol(ref)).all(), (f, k, err.max())


@pytest.mark.parametrize("binning", [0, 1])
@pytest.mark.parametrize("name,P,W,H,scale_mul", [("small", 3000, 200, 150, 0.8), ("big_gaussians", 800, 320, 240, 4.0),
                                                    ("tall", 2000, 48, 400, 1.0), ("wide", 2000, 400, 48, 1.0)])
def test_binning_schemes(name, P, W, H, scale_mul, binning, orc, scenes, rast, gpu):
    """Both binning schemes -- 0: column runs sorted by x, then one instance-level pass by tile row that expands the
    runs on the fly (default); 1: instance-level two-pass radix sort on tile ids -- must produce the reference's
    point_list and ranges bit for bit."""
    sc = scenes.synth(P, 77, scale_mul=scale_mul)
    cam = scenes.camera(1, 4, W, H)
    g = scenes.upstream_grad(H, W, 78)
    o32 = orc.render(sc, cam, g)
    o64 = orc.render(sc, cam, g, f64=True)
    rast._C.set_option("binning", binning)
    try:
        h = run_hip(rast, sc, cam, gpu, dL_dcolor=g)
    finally:
        rast._C.set_option("binning", 0)
    _check_forward_exact(o32, h)
    _check_grads(o64, o32, h, ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"], strict=True)


@pytest.mark.parametrize("name,P,W,H,scale_mul,aniso,opac_mul", [
    ("plain", 6000, 320, 240, 1.0, 1.0, 1.0),
    ("needles", 3000, 320, 240, 1.0, 60.0, 1.0),          # condition number of the conic up to ~1e5
    ("faint", 6000, 256, 192, 1.5, 1.0, 0.02),            # opacities around the 1/255 threshold
    ("huge", 300, 400, 304, 12.0, 4.0, 1.0),              # rectangles clamp to the whole grid
    ("ragged_edges", 4000, 203, 117, 1.0, 8.0, 0.5),      # last tile row / column partially outside the image
])
def test_tile_clipping_output_invariance(name, P, W, H, scale_mul, aniso, opac_mul, orc, scenes, rast, gpu):
    """tile_clip=1 (default) drops the tiles of a Gaussian's 3-sigma square that its alpha >= 1/255 ellipse cannot
    reach.  Outputs must not change by a single bit against the oracle (which walks the reference's literal lists),
    gradients stay within the bar, and the lists are ordered subsequences of the reference's."""
    sc = scenes.synth(P, 91, scale_mul=scale_mul)
    rng = np.random.default_rng(92)
    if aniso != 1.0:
        sc["scales"][:, 0] *= aniso ** rng.uniform(0.0, 1.0, size=P).astype(np.float32)
        sc["scales"][:, 1] /= aniso ** rng.uniform(0.0, 0.5, size=P).astype(np.float32)
    sc["opacities"] = (sc["opacities"] * opac_mul).astype(np.float32)
    cam = scenes.camera(1, 6, W, H)
    g = scenes.upstream_grad(H, W, 93)
    o32 = orc.render(sc, cam, g)
    o64 = orc.render(sc, cam, g, f64=True)
    h = run_hip(rast, sc, cam, gpu, dL_dcolor=g, tile_clip=1)
    _check_forward_exact(o32, h, clipped=True)
    _check_grads(o64, o32, h, ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"], strict=True,
                 conditioning=(aniso > 10.0))
    kept = int((h["ranges"].reshape(-1, 2)[:, 1].astype(np.int64) - h["ranges"].reshape(-1, 2)[:, 0]).sum())
    assert 0 < kept < o32["R"], "clipping should drop something on these scenes"


def test_backward_blend_per_pair_reduction_variant(orc, scenes, rast, gpu):
    """One pixel per lane has two backward blend kernels: the default blend_bwd_cull_t_kernel (cross-lane sums transposed out
    of the per-pair loop) and blend_bwd_cull_kernel<.., 1> (nine wave reductions per surviving pair; library switch
    "bwd_transposed" = 0).  Both meet the strict bar."""
    P, W, H = 6000, 200, 150
    sc = scenes.synth(P, 73, scale_mul=0.8)
    cam = scenes.camera(1, 5, W, H)
    g = scenes.upstream_grad(H, W, 74)
    o32 = orc.render(sc, cam, g)
    o64 = orc.render(sc, cam, g, f64=True)
    names = ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"]
    default = rast._C.get_option("bwd_transposed")
    for v in (0, 1):
        rast._C.set_option("bwd_transposed", v)
        try:
            h = run_hip(rast, sc, cam, gpu, dL_dcolor=g)
        finally:
            rast._C.set_option("bwd_transposed", default)
        _check_forward_exact(o32, h)
        _check_grads(o64, o32, h, names, strict=True)


@pytest.mark.parametrize("cull", [0, 1])
@pytest.mark.parametrize("ppl", [0, 1, 2, 4])
def test_pixels_per_lane_variants(ppl, cull, orc, scenes, rast, gpu):
    """The blend kernels exist in three work decompositions (1 / 2 / 4 pixels per lane = 4 / 2 / 1 waves
    per tile; 0 = picked from the tile count), each with and without wave-level strip culling; every
    combination must meet the same bars (culling must not change a single bit of the forward)."""
    P, W, H = 6000, 200, 150
    sc = scenes.synth(P, 71, scale_mul=0.8)
    sc["bg"] = np.array([0.2, 0.1, 0.4], np.float32)
    cam = scenes.camera(2, 5, W, H)
    g = scenes.upstream_grad(H, W, 72)
    o32 = orc.render(sc, cam, g)
    o64 = orc.render(sc, cam, g, f64=True)
    rast._C.set_option("pixels_per_lane", ppl)
    rast._C.set_option("cull", cull)
    try:
        h = run_hip(rast, sc, cam, gpu, dL_dcolor=g)
    finally:
        rast._C.set_option("pixels_per_lane", 0)
        rast._C.set_option("cull", 1)
    _check_forward_exact(o32, h)
    _check_grads(o64, o32, h, ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"], strict=True)


def test_grad_arena_zero_copy_bucket(scenes, rast, gpu):
    """Multi-GPU path: with a GradArena installed the backward writes the leaf gradients straight into
    one flat buffer (what view_parallel all-reduces in place).  Same numbers, and .grad aliases it."""
    import torch
    from conftest import settings_from
    P, W, H = 3000, 128, 96
    sc = scenes.synth(P, 81)
    cam = scenes.camera(1, 4, W, H)
    g = torch.as_tensor(scenes.upstream_grad(H, W, 82), device=gpu)
    rs = settings_from(rast, cam, sc, gpu)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)  # noqa: E731

    def run():
        leaves = {k: t(sc[k]).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
        m2 = torch.zeros((P, 3), device=gpu, requires_grad=True)
        color, _, _ = rast.GaussianRasterizer(rs)(means3D=leaves["means3D"], means2D=m2, opacities=leaves["opacities"],
                                                  shs=leaves["shs"], scales=leaves["scales"], rotations=leaves["rotations"])
        color.backward(g)
        leaves["means2D"] = m2
        return leaves

    plain = run()
    arena = rast._C.GradArena(P, 16, gpu)
    rast._C.set_grad_arena(arena)
    try:
        bucketed = run()
    finally:
        rast._C.set_grad_arena(None)
    assert arena.flat.numel() == P * 59
    lo, hi = arena.flat.data_ptr(), arena.flat.data_ptr() + arena.flat.numel() * 4
    for k in plain:
        a, b = plain[k].grad, bucketed[k].grad
        if k != "means2D":      # the screen-space gradient stays a private tensor (only its norm is exchanged)
            assert lo <= b.data_ptr() < hi, f"{k}.grad does not alias the arena (a copy was made)"
        tol = 1e-5 + 1e-4 * a.abs()
        assert ((a - b).abs() <= tol).all(), k      # two runs differ only by float-atomic ordering
    # the arena really is the concatenation of the gradients
    off = arena.offsets["sh"]
    assert torch.equal(arena.flat[off: off + P * 48].view(P, 16, 3), bucketed["shs"].grad)


def test_state_buffers_are_not_overrun(scenes, rast, gpu, monkeypatch):
    """Canary test: every state buffer the library asks for is allocated with a 4 KB tail of 0xA5; after forward +
    backward the tails must be intact (a write past the end of a buffer usually lands in the allocator's padding and
    goes unnoticed).  Sizes chosen to hit the small-input corners (P = 1, P <= 4096, tiny images, capacity-sized
    speculative launches after a big scene) and both binning schemes."""
    import torch
    from conftest import settings_from
    _C = rast._C
    PAD = 4096
    made = []

    def _make(self, slot):
        def alloc(_ctx, nbytes):
            buf = torch.empty(int(nbytes) + PAD, dtype=torch.uint8, device=self.device)
            buf[int(nbytes):] = 0xA5
            made.append((slot, int(nbytes), buf))
            self.buffers[slot] = buf
            return buf.data_ptr()
        return alloc

    monkeypatch.setattr(_C._Arena, "_make", _make)
    monkeypatch.setattr(_C, "PREALLOC_STATE", False)     # (all three buffers through the Python callbacks, so that each gets its canary tail)
    _C._tls.arena_pool = []          # (arenas are pooled per thread since round 6: the ones built before the patch keep their callbacks)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)  # noqa: E731
    seq = [(1, 64, 48), (7, 16, 16), (300, 97, 83), (5000, 320, 240), (1, 48, 32), (40, 640, 16), (4097, 128, 96), (2, 17, 5)]
    try:
        for binning in (0, 1):
            _C.set_option("binning", binning)
            for n, (P, W, H) in enumerate(seq):
                sc = scenes.synth(P, 131 + n)
                if P <= 2:
                    sc["means3D"][:] = 0.0
                cam = scenes.camera(n, 5, W, H)
                rs = settings_from(rast, cam, sc, gpu)
                leaves = {k: t(sc[k]).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
                m2 = torch.zeros((P, 3), device=gpu, requires_grad=True)
                made.clear()
                color, _, _ = rast.GaussianRasterizer(rs)(means3D=leaves["means3D"], means2D=m2, opacities=leaves["opacities"],
                                                          shs=leaves["shs"], scales=leaves["scales"], rotations=leaves["rotations"])
                color.backward(t(scenes.upstream_grad(H, W, 132)))
                torch.cuda.synchronize()
                assert len(made) >= 3
                for slot, nbytes, buf in made:
                    assert bool((buf[nbytes:] == 0xA5).all()), f"state buffer {slot} overrun (P={P}, {W}x{H}, binning={binning}, {nbytes} B)"
    finally:
        _C.set_option("binning", 0)
        _C._tls.arena_pool = []      # (the canary arenas must not serve the tests that follow)


@pytest.mark.remembered_cut_only
@pytest.mark.parametrize("speculative", [1, 0])
def test_speculative_launch_overflow_and_shrink(speculative, orc, scenes, rast, gpu):
    """The forward enqueues binning + blend against a capacity remembered from earlier calls, before it knows R and Q.
    A scene 60x larger than the previous one does not fit (the launch is repeated with exact sizes), a much smaller one
    runs inside an oversized capacity; both must be exact, as must the non-speculative path."""
    _C = rast._C
    _C.set_option("speculative", speculative)
    try:
        seq = [(300, 64, 48, 1.0), (20000, 320, 240, 1.0), (500, 96, 64, 0.5), (20000, 320, 240, 1.0)]
        redo0 = _C.get_option("redo_count")
        for n, (P, W, H, sm) in enumerate(seq):
            sc = scenes.synth(P, 111 + n, scale_mul=sm)
            cam = scenes.camera(n, 4, W, H)
            g = scenes.upstream_grad(H, W, 112)
            o32 = orc.render(sc, cam, g)
            o64 = orc.render(sc, cam, g, f64=True)
            for clip in (0, 1):
                h = run_hip(rast, sc, cam, gpu, dL_dcolor=g, tile_clip=clip)
                _check_forward_exact(o32, h, clipped=bool(clip))
                _check_grads(o64, o32, h, ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"], strict=True)
        redone = _C.get_option("redo_count") - redo0
        assert (redone >= 1) if speculative else (redone == 0)
    finally:
        _C.set_option("speculative", 1)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_gradient_factor_exchange(deg, scenes, rast, gpu):
    """Multi-GPU exchange of dL/dsh by its rank-1 factor (view_parallel.exchange_gradients, gsrast_sh_grad_combine):
    the batch mean of V views' dL/dsh recombined from V x 3 floats per Gaussian must equal the mean of the V dL/dsh
    tensors the plain backward writes (here the V views run in one process; the collective is a copy)."""
    import torch
    from conftest import settings_from
    import view_parallel
    P, W, H, V = 4000, 160, 120, 3
    sc = scenes.synth(P, 101, sh_degree=deg)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)  # noqa: E731
    g = t(scenes.upstream_grad(H, W, 102))
    _C = rast._C

    def run(k):
        cam = scenes.camera(k, V, W, H)
        rs = settings_from(rast, cam, sc, gpu)
        leaves = {n: t(sc[n]).requires_grad_(True) for n in ("means3D", "shs", "opacities", "scales", "rotations")}
        m2 = torch.zeros((P, 3), device=gpu, requires_grad=True)
        color, _, _ = rast.GaussianRasterizer(rs)(means3D=leaves["means3D"], means2D=m2, opacities=leaves["opacities"],
                                                  shs=leaves["shs"], scales=leaves["scales"], rotations=leaves["rotations"])
        color.backward(g)
        return leaves

    plain = [run(k) for k in range(V)]
    want = {n: sum(p[n].grad for p in plain) / V for n in plain[0]}

    # one "rank" per view: every backward leaves its factor chunk; the all-gather is emulated by stacking the chunks
    arena = _C.GradArena(P, 16, gpu, sh_factors=True, world=1)
    _C.set_grad_arena(arena)
    try:
        chunks, dense = [], torch.zeros_like(arena.dense)
        for k in range(V):
            arena.zero_grad()           # every "rank" starts its own step
            lv = run(k)
            chunks.append(arena.factor.clone())
            dense += arena.dense
            if k == 0:      # single view through the public entry point: shs.grad (a view of the arena) gets filled
                view_parallel.exchange_gradients(arena, lv["means3D"].detach(), 1)
                a, b = plain[0]["shs"].grad, lv["shs"].grad
                assert ((a - b).abs() <= 1e-6 + 1e-4 * a.abs()).all()      # two runs: float-atomic order upstream
        got_sh = _C.sh_grad_combine(arena, lv["means3D"].detach(), torch.cat(chunks), V, 1.0 / V).clone()
    finally:
        _C.set_grad_arena(None)
    a = want["shs"]
    assert ((a - got_sh).abs() <= 1e-6 + 1e-4 * a.abs()).all(), float((a - got_sh).abs().max())
    assert not got_sh[:, (deg + 1) ** 2:, :].any()
    # the dense part is the concatenation means3D | opacity | scales | rotations
    dense /= V
    o = 0
    for n, w in (("means3D", 3), ("opacities", 1), ("scales", 3), ("rotations", 4)):
        a, b = want[n].reshape(-1), dense[o: o + P * w]
        assert ((a - b).abs() <= 1e-5 + 1e-4 * a.abs()).all(), n
        o += P * w


@pytest.mark.parametrize("M,deg", [(1, 0), (4, 1), (9, 2), (16, 1), (25, 3)])
def test_sh_row_lengths(M, deg, orc, scenes, rast, gpu):
    """max_coeffs M other than 16 (M*3 floats per row: 3, 12, 27, 48, 75 -- aligned and unaligned rows,
    staged through LDS up to 48 floats and read directly beyond), active degree below what M allows."""
    P, W, H = 1500, 112, 80
    sc = scenes.synth(P, 90 + M, sh_degree=deg)
    rng = np.random.default_rng(M)
    shs = np.zeros((P, M, 3), np.float32)
    shs[:, 0] = rng.uniform(-1.7, 1.7, size=(P, 3))
    shs[:, 1:] = rng.normal(0, 0.2, size=(P, M - 1, 3))
    sc["shs"] = shs
    cam = scenes.camera(1, 3, W, H)
    g = scenes.upstream_grad(H, W, 5)
    o32 = orc.render(sc, cam, g)
    o64 = orc.render(sc, cam, g, f64=True)
    h = run_hip(rast, sc, cam, gpu, dL_dcolor=g)
    _check_forward_exact(o32, h)
    _check_grads(o64, o32, h, ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"], strict=True)
    ncoef = (deg + 1) ** 2
    assert not h["dL_dsh"][:, ncoef:].any()          # coefficients above the active degree get zero gradient


def test_scale_modifier_single_gaussian_and_views(orc, scenes, rast, gpu):
    """scale_modifier != 1 (settings field the reference multiplies into the scales, forward.cu:122-124);
    P = 1; non-contiguous input views (the binding makes them contiguous like the reference's .contiguous())."""
    import torch
    from conftest import settings_from
    W, H = 80, 64
    sc = scenes.synth(700, 95)
    cam = scenes.camera(0, 1, W, H)
    cam["scale_modifier"] = 0.6
    g = scenes.upstream_grad(H, W, 6)
    o32 = orc.render(sc, cam, g)
    o64 = orc.render(sc, cam, g, f64=True)
    h = run_hip(rast, sc, cam, gpu, dL_dcolor=g)
    _check_forward_exact(o32, h)
    _check_grads(o64, o32, h, ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"], strict=True)
    # one Gaussian in the middle of the screen
    one = {k: (v[:1].copy() if isinstance(v, np.ndarray) and v.shape[:1] == (700,) else v) for k, v in sc.items()}
    one["means3D"][:] = 0.0
    cam1 = scenes.camera(0, 1, W, H)
    o1 = orc.render(one, cam1, g)
    h1 = run_hip(rast, one, cam1, gpu, dL_dcolor=g)
    _check_forward_exact(o1, h1)
    assert h1["R"] == o1["R"] > 0
    # strided views of larger tensors
    rs = settings_from(rast, cam1, sc, gpu)
    big = torch.as_tensor(np.repeat(sc["means3D"], 2, axis=0), device=gpu)
    means_view = big[::2]                                       # stride (6, 1): not contiguous
    assert not means_view.is_contiguous()
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)  # noqa: E731
    shs_view = torch.as_tensor(np.ascontiguousarray(sc["shs"].transpose(1, 0, 2)), device=gpu).permute(1, 0, 2)
    assert not shs_view.is_contiguous()
    color, radii, _ = rast.GaussianRasterizer(rs)(means3D=means_view, means2D=torch.zeros(700, 3, device=gpu), opacities=t(sc["opacities"]),
                                                  shs=shs_view, scales=t(sc["scales"]), rotations=t(sc["rotations"]))
    ref = orc.render(sc, cam1, None)
    np.testing.assert_array_equal(bits(color.cpu().numpy()), bits(ref["out_color"]))


def test_runs_on_a_side_stream(orc, scenes, rast, gpu):
    """Kernels are enqueued on torch's CURRENT stream (the reference used the legacy default stream)."""
    import torch
    P, W, H = 2000, 96, 96
    sc = scenes.synth(P, 97)
    cam = scenes.camera(2, 6, W, H)
    g = scenes.upstream_grad(H, W, 7)
    o32 = orc.render(sc, cam, g)
    s = torch.cuda.Stream(device=gpu)
    with torch.cuda.stream(s):
        h = run_hip(rast, sc, cam, gpu, dL_dcolor=g)
    s.synchronize()
    _check_forward_exact(o32, h)


@pytest.mark.parametrize("W,H", [(16, 16), (1, 1), (17, 5), (300, 8)])
def test_degenerate_image_sizes(W, H, orc, scenes, rast, gpu):
    """One tile, one pixel, a partial tile, a one-tile-high strip."""
    P = 300
    sc = scenes.synth(P, 111, scale_mul=0.5)
    cam = scenes.camera(0, 1, W, H)
    g = scenes.upstream_grad(H, W, 3)
    o32 = orc.render(sc, cam, g)
    o64 = orc.render(sc, cam, g, f64=True)
    h = run_hip(rast, sc, cam, gpu, dL_dcolor=g)
    _check_forward_exact(o32, h)
    _check_grads(o64, o32, h, ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"])


def test_cov3d_against_reference_python_vectors(scenes, rast, gpu):
    """The HIP kernel's cov3D (state export) against the reference's OWN Python build_covariance_from_scaling_rotation
    (utils/general_utils.py:113-205 composed as scene/saro_gaussian.py:33-37; vectors generated by importing the reference,
    tests/golden/make_golden.py): packing order, quaternion convention, Sigma = R S S^T R^T, scale_modifier."""
    import os
    from test_oracle_golden import G, _cov_scene
    z = np.load(os.path.join(G, "ref_python_vectors.npz"))
    sc, cam = _cov_scene(scenes, z)
    for tag, mod in (("1", 1.0), ("0p7", 0.7)):
        cam["scale_modifier"] = mod
        h = run_hip(rast, sc, cam, gpu)
        want = z["cov3D_mod" + tag].astype(np.float64)
        vis = h["radii"] > 0
        assert vis.mean() > 0.9
        tol = 2e-6 * np.abs(want).max(axis=1, keepdims=True)
        assert (np.abs(h["cov3D"] - want)[vis] <= tol[vis]).all()


def test_grad_arena_contract(scenes, rast, gpu):
    """GradArena: P not a multiple of 4 (every segment still starts 16-byte aligned: dL/drot leaves through float4 stores); a second
    backward of the same step ADDS (the reference's batch loop, cache_gradient); factor mode refuses what it cannot do -- a second
    backward before zero_grad(), or an `shs` that is not a leaf (cat(features_dc, features_rest))."""
    import torch
    from conftest import settings_from
    P, W, H = 3001, 128, 96
    sc = scenes.synth(P, 83)
    g = torch.as_tensor(scenes.upstream_grad(H, W, 84), device=gpu)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)  # noqa: E731
    _C = rast._C

    def render(leaves, k, shs=None):
        cam = scenes.camera(k, 4, W, H)
        rs = settings_from(rast, cam, sc, gpu)
        m2 = torch.zeros((P, 3), device=gpu, requires_grad=True)
        color, _, _ = rast.GaussianRasterizer(rs)(means3D=leaves["means3D"], means2D=m2, opacities=leaves["opacities"],
                                                  shs=leaves["shs"] if shs is None else shs, scales=leaves["scales"], rotations=leaves["rotations"])
        return color

    def fresh():
        return {k: t(sc[k]).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}

    plain = fresh()
    for k in (0, 1):
        render(plain, k).backward(g)                 # autograd sums the two views' gradients
    arena = _C.GradArena(P, 16, gpu)
    assert all(o % 4 == 0 for o in arena.offsets.values()) and arena.flat.data_ptr() % 16 == 0
    _C.set_grad_arena(arena)
    try:
        lv = fresh()
        arena.zero_grad()
        for k in (0, 1):
            render(lv, k).backward(g)
        lo, hi = arena.flat.data_ptr(), arena.flat.data_ptr() + arena.flat.numel() * 4
        for n in plain:
            a, b = plain[n].grad, lv[n].grad
            assert lo <= b.data_ptr() < hi, f"{n}.grad left the arena"
            assert ((a - b).abs() <= 1e-6 + 1e-4 * a.abs()).all(), n
    finally:
        _C.set_grad_arena(None)
    fa = _C.GradArena(P, 16, gpu, sh_factors=True, world=1)
    _C.set_grad_arena(fa)
    try:
        lv = fresh()
        fa.zero_grad()
        render(lv, 0).backward(g)
        with pytest.raises(RuntimeError, match="second backward"):
            render(lv, 1).backward(g)
        fa.zero_grad()
        dc, rest = lv["shs"].detach()[:, :1].clone().requires_grad_(True), lv["shs"].detach()[:, 1:].clone().requires_grad_(True)
        with pytest.raises(RuntimeError, match="leaf"):
            render(lv, 0, shs=torch.cat([dc, rest], dim=1))
    finally:
        _C.set_grad_arena(None)


def test_two_threads_two_streams_different_options(orc, scenes, rast, gpu):
    """The C ABI is reentrant (include/gsrast.h, gsrast_forward_ex / gsrast_backward_ex): two host threads on two streams, each
    with its OWN options (thread A: the reference's literal lists + instance-level binning + 2 pixels per lane in the backward;
    thread B: product defaults), interleaved for many iterations, reproduce their single-threaded results bit for bit (forward)
    and within the float-atomic band (backward)."""
    import threading
    import torch
    from conftest import settings_from
    _C = rast._C
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)  # noqa: E731
    jobs = {"A": dict(P=6000, W=320, H=208, seed=91, opts=dict(tile_clip=0, binning=1, bwd_pixels_per_lane=2, speculative=0)),
            "B": dict(P=9000, W=256, H=256, seed=92, opts=dict())}

    def run(job, n_iter, out, opts_after=None):
        for k, v in job["opts"].items():
            _C.set_option(k, v)                       # per-thread
        sc = scenes.synth(job["P"], job["seed"])
        cam = scenes.camera(1, 3, job["W"], job["H"])
        rs = settings_from(rast, cam, sc, gpu)
        g = t(scenes.upstream_grad(job["H"], job["W"], job["seed"] + 1))
        stream = torch.cuda.Stream(device=gpu)
        res = []
        with torch.cuda.stream(stream):
            for _ in range(n_iter):
                lv = {k: t(sc[k]).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
                m2 = torch.zeros((job["P"], 3), device=gpu, requires_grad=True)
                color, radii, depth = rast.GaussianRasterizer(rs)(means3D=lv["means3D"], means2D=m2, opacities=lv["opacities"], shs=lv["shs"],
                                                                  scales=lv["scales"], rotations=lv["rotations"])
                color.backward(g)
                res.append((color.detach().clone(), depth.clone(), radii.clone(), {k: v.grad.clone() for k, v in lv.items()}, _C.get_option("last_instances")))
            stream.synchronize()
        out.extend(res)
        if opts_after is not None:
            opts_after.update({k: _C.get_option(k) for k in ("tile_clip", "binning", "bwd_pixels_per_lane")})

    single = {}
    for name, job in jobs.items():                    # single-threaded references, each in a fresh thread (fresh per-thread options)
        out = []
        th = threading.Thread(target=run, args=(job, 1, out)); th.start(); th.join()
        single[name] = out[0]
    outs, seen = {"A": [], "B": []}, {"A": {}, "B": {}}
    ths = [threading.Thread(target=run, args=(jobs[n], 12, outs[n], seen[n])) for n in ("A", "B")]
    for th in ths: th.start()
    for th in ths: th.join()
    assert seen["A"] == dict(tile_clip=0, binning=1, bwd_pixels_per_lane=2) and seen["B"] == dict(tile_clip=1, binning=0, bwd_pixels_per_lane=0)
    assert _C.get_option("tile_clip") == 1 and _C.get_option("binning") == 0        # the main thread's options were never touched
    for n in ("A", "B"):
        c0, d0, r0, g0, R0 = single[n]
        assert len(outs[n]) == 12
        for c, d, r, gr, R in outs[n]:
            assert torch.equal(c, c0) and torch.equal(d, d0) and torch.equal(r, r0) and R == R0
            for k in g0:
                assert ((gr[k] - g0[k]).abs() <= 1e-6 + 1e-4 * g0[k].abs()).all(), (n, k)


def test_second_backward_on_the_same_forward_state(scenes, rast, gpu):
    """retain_graph: the forward leaves the per-Gaussian gradient records zero and the FIRST backward skips their zero-fill
    (options.grads_zeroed); a second backward on the same state must fill them again -- both give the same gradients."""
    import torch
    from conftest import settings_from
    P, W, H = 5000, 160, 128
    sc = scenes.synth(P, 191)
    cam = scenes.camera(1, 4, W, H)
    rs = settings_from(rast, cam, sc, gpu)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)  # noqa: E731
    g = t(scenes.upstream_grad(H, W, 192))
    lv = {k: t(sc[k]).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    m2 = torch.zeros((P, 3), device=gpu, requires_grad=True)
    color, _, _ = rast.GaussianRasterizer(rs)(means3D=lv["means3D"], means2D=m2, opacities=lv["opacities"], shs=lv["shs"],
                                              scales=lv["scales"], rotations=lv["rotations"])
    color.backward(g, retain_graph=True)
    first = {k: v.grad.clone() for k, v in lv.items()}
    first["m2"] = m2.grad.clone()
    for v in list(lv.values()) + [m2]:
        v.grad = None
    color.backward(g)
    for k, v in list(lv.items()) + [("m2", m2)]:
        a, b = first[k], v.grad
        assert float(a.abs().max()) > 0
        assert ((a - b).abs() <= 1e-7 + 1e-4 * a.abs()).all(), k


def test_two_phase_backward_factor_ready_hook(scenes, rast, gpu):
    """options.backward_phase: phase 1 (blend backward) leaves the view's dL/dsh factor final -- what a multi-GPU caller needs to
    start its all-gather (view_parallel.overlap_factor_exchange) -- and phase 2 (per-Gaussian backward) the rest; 1 then 2 is the
    one-call backward.  Through the Python hook: the factor seen by the hook is the factor after the whole backward, and every
    gradient equals the hook-less run's (float-atomic order apart)."""
    import torch
    from conftest import settings_from
    _C = rast._C
    P, W, H, deg = 5000, 200, 144, 3
    sc = scenes.synth(P, 171, sh_degree=deg, scale_mul=1.3)
    sc["shs"][::7, 0, :] = -3.0        # some clamped colour channels: their factor is masked
    cam = scenes.camera(1, 4, W, H)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)  # noqa: E731
    g = t(scenes.upstream_grad(H, W, 172))
    rs = settings_from(rast, cam, sc, gpu)

    def run():
        leaves = {n: t(sc[n]).requires_grad_(True) for n in ("means3D", "shs", "opacities", "scales", "rotations")}
        m2 = torch.zeros((P, 3), device=gpu, requires_grad=True)
        color, radii, _ = rast.GaussianRasterizer(rs)(means3D=leaves["means3D"], means2D=m2, opacities=leaves["opacities"],
                                                      shs=leaves["shs"], scales=leaves["scales"], rotations=leaves["rotations"])
        color.backward(g)
        torch.cuda.synchronize()
        return leaves, m2, radii

    arena = _C.GradArena(P, 16, gpu, sh_factors=True, world=1)
    _C.set_grad_arena(arena)
    seen = []
    try:
        arena.zero_grad()
        run()
        one_call = {"factor": arena.factor.clone(), "dense": arena.dense.clone()}
        _C.set_factor_ready_hook(lambda ar: seen.append((ar.factor.clone(), ar.dense.clone())))
        arena.zero_grad()
        arena.dense.fill_(float("nan"))            # phase 2 writes every dense gradient
        lv, m2, radii = run()
    finally:
        _C.set_factor_ready_hook(None)
        _C.set_grad_arena(None)
    assert len(seen) == 1
    at_hook, dense_at_hook = seen[0]
    assert torch.equal(at_hook, arena.factor), "the factor must be final when the hook runs"
    assert torch.isnan(dense_at_hook).all(), "phase 1 must not touch the per-Gaussian outputs"
    assert not torch.isnan(arena.dense).any() and not torch.isnan(m2.grad).any()
    for a, b, what in ((one_call["factor"], arena.factor, "factor"), (one_call["dense"], arena.dense, "dense")):
        assert ((a - b).abs() <= 1e-6 + 1e-4 * a.abs()).all(), what
    fac = arena.factor[: 3 * P].reshape(P, 3)
    assert fac.abs().max() > 0 and not fac[radii == 0].any()


@pytest.mark.parametrize("deg", [1, 2, 3])
def test_direction_derivatives_from_the_forward_equal_the_backward_kernel(deg, scenes, rast, gpu):
    """Round 3: the forward's colour kernel stores d(colour)/d(view direction) (backward.cu:78-127) for the backward while the SH
    block is in LDS.  options.forward_only = 1 on both calls selects round 2's route (the backward re-reads the SH blocks in
    sh_dir_derivs_kernel): same expressions on the same operands -- the gradients agree to the run-to-run noise of the blend
    backward's float atomics (the bar of test_second_backward_on_the_same_forward_state)."""
    import torch
    from conftest import settings_from
    _C = rast._C
    P, W, H = 6000, 176, 128
    sc = scenes.synth(P, 301 + deg, sh_degree=deg)
    cam = scenes.camera(2, 5, W, H)
    rs = settings_from(rast, cam, sc, gpu)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)  # noqa: E731
    g = t(scenes.upstream_grad(H, W, 302))
    e = torch.empty(0)
    res = []
    for fo in (0, 1):
        _C.set_option("forward_only", fo)
        try:
            R, color, radii, gb, bb, ib, depth = _C.rasterize_gaussians(
                rs.bg, t(sc["means3D"]), e, t(sc["opacities"]), t(sc["scales"]), t(sc["rotations"]), 1.0, e, rs.viewmatrix, rs.projmatrix,
                rs.tanfovx, rs.tanfovy, H, W, t(sc["shs"]), deg, rs.campos, False)
            grads = _C.rasterize_gaussians_backward(
                rs.bg, t(sc["means3D"]), radii, e, t(sc["scales"]), t(sc["rotations"]), 1.0, e, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
                rs.tanfovy, g, t(sc["shs"]), deg, rs.campos, gb, R, bb, ib, first_backward=True)
        finally:
            _C.set_option("forward_only", 0)
        torch.cuda.synchronize()
        res.append((color, [x.clone() for x in grads]))
    assert torch.equal(res[0][0], res[1][0])
    assert float(res[0][1][3].abs().max()) > 0
    for a, b in zip(res[0][1], res[1][1]):
        assert ((a - b).abs() <= 1e-7 + 1e-4 * a.abs()).all()


def test_no_grad_forward_skips_the_backward_preparation(orc, scenes, rast, gpu):
    """Evaluation (torch.no_grad(), or no input requiring a gradient): the autograd node tells the library that no backward follows
    (options.forward_only); outputs are bit-identical to the training-mode forward."""
    import torch
    from conftest import settings_from
    P, W, H = 4000, 160, 112
    sc = scenes.synth(P, 311)
    cam = scenes.camera(0, 3, W, H)
    rs = settings_from(rast, cam, sc, gpu)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)  # noqa: E731
    kw = dict(means3D=t(sc["means3D"]), means2D=torch.zeros((P, 3), device=gpu), opacities=t(sc["opacities"]), shs=t(sc["shs"]),
              scales=t(sc["scales"]), rotations=t(sc["rotations"]))
    with torch.no_grad():
        c0, r0, d0 = rast.GaussianRasterizer(rs)(**kw)
    kw["means3D"] = kw["means3D"].clone().requires_grad_(True)
    c1, r1, d1 = rast.GaussianRasterizer(rs)(**kw)
    assert torch.equal(c0, c1.detach()) and torch.equal(r0, r1) and torch.equal(d0, d1)
    o = orc.render(sc, cam, None)
    assert np.array_equal(bits(c0.cpu().numpy()), bits(o["out_color"]))


def test_launch_order_hints_never_change_a_result(orc, scenes, rast, gpu):
    """The context remembers, per camera pose, how deep every tile's list was consumed and orders the next forward blend of that pose
    by it (include/gsrast.h: options.no_order_hint).  Only the launch order may depend on it: alternating poses, more poses than the
    table has slots (least-recently-used replacement), another image size in between, and hints switched off all give bit-identical
    outputs and state -- and the first render of every pose equals the oracle."""
    import torch
    from conftest import settings_from
    _C = rast._C
    P = 40_000                      # the work-bucket launch order (and with it the hints) is in force on the run-compressed path
    sc = scenes.synth(P, 601, scale_mul=0.7)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)  # noqa: E731
    ten = {k: t(sc[k]) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    e = torch.empty(0)

    def render(cam, W, H):
        rs = settings_from(rast, cam, sc, gpu)
        R, color, radii, gb, bb, ib, depth = _C.rasterize_gaussians(
            rs.bg, ten["means3D"], e, ten["opacities"], ten["scales"], ten["rotations"], 1.0, e, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
            rs.tanfovy, H, W, ten["shs"], 3, rs.campos, False)
        st = _C.debug_export(P, R, W, H, gb, bb, ib)
        return R, color.clone(), depth.clone(), radii.clone(), st["n_contrib"].clone(), st["final_T"].clone()

    def same(a, b):
        return a[0] == b[0] and all(torch.equal(x, y) for x, y in zip(a[1:], b[1:]))

    W, H = 320, 240
    NP = 264                                                         # poses > the table's 256 slots
    cams = [scenes.camera(k, NP, W, H) for k in range(NP)]
    first = {}
    _C.set_option("list_cut_always", 1)                              # the list cut rides on the same table: exercised here as well
    for k in (0, 1, 0, 1, 0):                                        # alternating: the third render of pose 0 runs by its own hint
        out = render(cams[k], W, H)
        if k in first:
            assert same(out, first[k]), f"pose {k}: a hinted launch order changed a result"
        first[k] = out
    o = orc.render(sc, cams[0])
    assert first[0][0] == o["R"] and np.array_equal(bits(first[0][1].cpu().numpy()), bits(o["out_color"]))
    for k in range(2, NP):                                           # overflow the table: pose 0 and 1 are evicted
        first[k] = render(cams[k], W, H)
    other = render(scenes.camera(3, 7, 200, 152), 200, 152)          # another image size: the context starts a new table
    assert same(render(scenes.camera(3, 7, 200, 152), 200, 152), other)
    for k in (0, 1, NP - 1, 20):
        assert same(render(cams[k], W, H), first[k])
        assert same(render(cams[k], W, H), first[k])
    _C.set_option("no_order_hint", 1)
    try:
        for k in (0, NP - 1):
            assert same(render(cams[k], W, H), first[k])
    finally:
        _C.set_option("no_order_hint", 0)
        _C.set_option("list_cut_always", 0)


@pytest.mark.remembered_cut_only
def test_list_cut_is_verified_and_never_changes_a_result(orc, scenes, rast, gpu):
    """List cut (include/gsrast.h: options.no_list_cut): the second forward of a pose gives column runs only to the Gaussians in front
    of the cut depth of some tile they cover.  The speculation is verified on the device: same outputs and state bit for bit, the same
    gradients; a scene that turned transparent behind the context's back (the cut lists are too short) is redone from the full lists
    inside the same call (cut_fallbacks counts it) and equals the oracle; the next forward has learned the new cut depths."""
    import torch
    from conftest import settings_from
    _C = rast._C
    P, W, H = 60_000, 320, 240
    sc = scenes.synth(P, 811, scale_mul=1.3)          # dense enough for most tiles to saturate early
    cam = scenes.camera(2, 9, W, H)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)  # noqa: E731
    e = torch.empty(0)

    def render(scene):
        rs = settings_from(rast, cam, scene, gpu)
        ten = {k: t(scene[k]) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
        R, color, radii, gb, bb, ib, depth = _C.rasterize_gaussians(
            rs.bg, ten["means3D"], e, ten["opacities"], ten["scales"], ten["rotations"], 1.0, e, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
            rs.tanfovy, H, W, ten["shs"], 3, rs.campos, False)
        st = _C.debug_export(P, R, W, H, gb, bb, ib)
        return (R, color.clone(), depth.clone(), radii.clone(), st["n_contrib"].clone(), st["final_T"].clone()), _C.context_query("last_late")

    def same(a, b):
        return a[0] == b[0] and all(torch.equal(x, y) for x, y in zip(a[1:], b[1:]))

    _C.set_option("list_cut_always", 1)               # (by default the cut is only applied where it pays: scenes of millions of column runs)
    _C.set_option("near_pose", 0)                     # (a first render must find nothing to cut by: no borrowing from whatever pose another test left near this one)
    try:
        _list_cut_body(orc, scenes, rast, gpu, _C, render, same, sc, cam, P, W, H)
        # option "layer_cut" (off by default, DESIGN.md: measured slower): a pose WITHOUT remembered cut depths lists the nearest
        # eighth of the Gaussians first and completes the tiles that did not saturate inside it -- same results, another pose
        cam2 = scenes.camera(5, 9, W, H)

        def render2(scene):
            nonlocal cam
            keep, cam = cam, cam2
            try:
                return render(scene)
            finally:
                cam = keep
        _C.set_option("layer_cut", 1)
        try:
            _list_cut_body(orc, scenes, rast, gpu, _C, render2, same, sc, cam2, P, W, H, layer=True)
        finally:
            _C.set_option("layer_cut", 0)
        # option "chain_gate" off: the completion pass's launches on the caller's stream (round 3's arrangement) instead of behind the gate
        # on the context's second stream -- same results, a third pose
        cam3 = scenes.camera(7, 9, W, H)

        def render3(scene):
            nonlocal cam
            keep, cam = cam, cam3
            try:
                return render(scene)
            finally:
                cam = keep
        _C.set_option("chain_gate", 0)
        try:
            _list_cut_body(orc, scenes, rast, gpu, _C, render3, same, sc, cam3, P, W, H)
        finally:
            _C.set_option("chain_gate", 1)
    finally:
        _C.set_option("list_cut_always", 0)
        _C.set_option("near_pose", 0)


@pytest.mark.remembered_cut_only
def test_near_pose_borrows_cut_depths_and_never_changes_a_result(orc, scenes, rast, gpu):
    """A pose the context's table does not know takes the launch order and the cut depths (widened over 7 x 7 tiles) of a NEAR pose's slot
    (option near_pose, gsrast_common.h HintTable::cam): along a camera path every frame after the first two is cut although no pose is
    ever rendered twice -- with the same outputs bit for bit as without any cut, and as the oracle."""
    import torch
    from conftest import settings_from
    _C = rast._C
    P, W, H, V = 60_000, 336, 256, 360               # (an image size no other test uses: this test's poses are the only ones in its table)
    sc = scenes.synth(P, 811, scale_mul=1.3)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)  # noqa: E731
    e = torch.empty(0)
    ten = {k: t(sc[k]) for k in ("means3D", "opacities", "shs", "scales", "rotations")}

    def render(k):
        cam = scenes.camera(k, V, W, H)
        rs = settings_from(rast, cam, sc, gpu)
        R, color, radii, gb, bb, ib, depth = _C.rasterize_gaussians(
            rs.bg, ten["means3D"], e, ten["opacities"], ten["scales"], ten["rotations"], 1.0, e, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
            rs.tanfovy, H, W, ten["shs"], 3, rs.campos, False)
        st = _C.debug_export(P, R, W, H, gb, bb, ib)
        return (R, color.clone(), depth.clone(), radii.clone(), st["n_contrib"].clone(), st["final_T"].clone()), _C.context_query("last_late")

    same = lambda a, b: a[0] == b[0] and all(torch.equal(x, y) for x, y in zip(a[1:], b[1:]))  # noqa: E731
    frames2 = list(range(200, 212))                   # one degree apart, no pose twice
    _C.set_option("list_cut_always", 1)
    _C.set_option("near_pose", 3)                     # (off by default since round 5: a pose without a slot gets PREDICTED cut depths)
    try:
        cut_frames = 0
        outs = {}
        for k in frames2:
            outs[k], late = render(k)
            cut_frames += 1 if late > 0 else 0
        assert cut_frames >= len(frames2) - 2, cut_frames            # the first frame has nobody to borrow from, the second only a first estimate
        _C.set_option("no_list_cut", 1)
        try:
            for k in frames2:
                assert same(outs[k], render(k)[0]), k
        finally:
            _C.set_option("no_list_cut", 0)
        k = frames2[-1]
        o = orc.render(sc, scenes.camera(k, V, W, H))
        assert outs[k][0] == o["R"] and np.array_equal(bits(outs[k][1].cpu().numpy()), bits(o["out_color"]))
    finally:
        _C.set_option("list_cut_always", 0)
        _C.set_option("near_pose", 0)


def _list_cut_body(orc, scenes, rast, gpu, _C, render, same, sc, cam, P, W, H, layer=False):
    fb0 = _C.context_query("cut_fallbacks")
    full, late0 = render(sc)                          # first render of the pose by this context: no remembered cut depths
    if layer:                                         # option "layer_cut": a depth LAYER is listed first and completed behind the blend
        assert late0 > 0
        fb0 = _C.context_query("cut_fallbacks")       # (completion passes so far)
    else:
        assert late0 == 0                             # nothing to cut by
    o = orc.render(sc, cam)
    assert full[0] == o["R"] and np.array_equal(bits(full[1].cpu().numpy()), bits(o["out_color"]))
    cut1, late1 = render(sc)
    cut2, late2 = render(sc)
    assert late1 > P // 4 and late2 > P // 4, (late1, late2)         # the cube is opaque after a fraction of its depth
    assert same(cut1, full) and same(cut2, full)
    assert _C.context_query("cut_fallbacks") == fb0                  # the speculation held: nothing was redone
    _C.set_option("no_list_cut", 1)
    try:
        off, late_off = render(sc)
    finally:
        _C.set_option("no_list_cut", 0)
    assert late_off == 0 and same(off, full)

    # the scene turns transparent: every tile now consumes far more than 1.5 x what it did
    sc2 = dict(sc)
    sc2["opacities"] = (sc["opacities"] * 0.04).astype(np.float32)
    o2 = orc.render(sc2, cam)
    thin, late3 = render(sc2)
    assert late3 > 0                                                  # the cut was applied ...
    assert _C.context_query("cut_fallbacks") == fb0 + 1               # ... found too short, and everything was redone from the full lists
    assert thin[0] == o2["R"] and np.array_equal(bits(thin[1].cpu().numpy()), bits(o2["out_color"]))
    assert np.array_equal(bits(thin[2].cpu().numpy()), bits(o2["out_depth"]))
    thin2, _ = render(sc2)
    assert same(thin2, thin)
    assert _C.context_query("cut_fallbacks") == fb0 + 1               # the redo left cut depths that fit the new scene (or none)

    # gradients through a cut forward: the autograd path, oracle bar
    g = scenes.upstream_grad(H, W, 812) * (H * W)
    o32 = orc.render(sc, cam, g)
    o64 = orc.render(sc, cam, g, f64=True)
    run_hip(rast, sc, cam, gpu, dL_dcolor=g, tile_clip=1)            # (re-learns the opaque scene's cut depths; may fall back once)
    h = run_hip(rast, sc, cam, gpu, dL_dcolor=g, tile_clip=1)
    assert _C.context_query("last_late") > P // 4
    assert np.array_equal(bits(h["out_color"]), bits(o32["out_color"]))
    _check_grads(o64, o32, h, ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"])


@pytest.mark.remembered_cut_only
def test_list_cut_on_a_large_image(scenes, rast, gpu):
    """An image of more than 3072 cells of 2 x 2 tiles (here 2048 x 1600: 128 x 100 tiles) keeps its cut depths in 4 x 4-tile cells
    (gsrast_common.h: cut_cell_shift): same results with and without the cut, also after the scene turned transparent."""
    import torch
    from conftest import settings_from
    _C = rast._C
    P, W, H = 80_000, 2048, 1600
    sc = scenes.synth(P, 921, scale_mul=1.5)
    cam = scenes.camera(1, 5, W, H)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)  # noqa: E731
    e = torch.empty(0)

    def render(scene):
        rs = settings_from(rast, cam, scene, gpu)
        ten = {k: t(scene[k]) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
        R, color, radii, gb, bb, ib, depth = _C.rasterize_gaussians(
            rs.bg, ten["means3D"], e, ten["opacities"], ten["scales"], ten["rotations"], 1.0, e, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
            rs.tanfovy, H, W, ten["shs"], 3, rs.campos, False)
        st = _C.debug_export(P, R, W, H, gb, bb, ib)
        return (R, color.clone(), depth.clone(), radii.clone(), st["n_contrib"].clone(), st["final_T"].clone()), _C.context_query("last_late")

    def same(a, b):
        return a[0] == b[0] and all(torch.equal(x, y) for x, y in zip(a[1:], b[1:]))

    _C.set_option("list_cut_always", 1)
    try:
        _C.set_option("no_list_cut", 1)
        try:
            ref, _ = render(sc)
        finally:
            _C.set_option("no_list_cut", 0)
        a, late_a = render(sc)
        b, late_b = render(sc)
        assert late_a > 1000 and late_b > 1000, (late_a, late_b)        # (a sparse scene at this size: few tiles saturate, few Gaussians are late)
        assert same(a, ref) and same(b, ref)
        fb0 = _C.context_query("cut_fallbacks")
        sc2 = dict(sc)
        sc2["opacities"] = (sc["opacities"] * 0.03).astype(np.float32)
        thin, late_c = render(sc2)
        assert late_c > 0 and _C.context_query("cut_fallbacks") == fb0 + 1
        _C.set_option("no_list_cut", 1)
        try:
            ref2, _ = render(sc2)
        finally:
            _C.set_option("no_list_cut", 0)
        assert same(thin, ref2)
    finally:
        _C.set_option("list_cut_always", 0)


@pytest.mark.remembered_cut_only
def test_list_cut_survives_alternating_image_sizes(scenes, rast, gpu):
    """a context keeps one pose table per image size (up to four): train and eval resolutions that alternate call by
    call each keep their poses' cut depths -- the cut is in force on every visit after a size's first, with the same results as without."""
    import torch
    from conftest import settings_from
    _C = rast._C
    P = 60_000
    sc = scenes.synth(P, 811, scale_mul=1.3)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)  # noqa: E731
    e = torch.empty(0)
    ten = {k: t(sc[k]) for k in ("means3D", "opacities", "shs", "scales", "rotations")}

    def render(W, H):
        cam = scenes.camera(2, 9, W, H)
        rs = settings_from(rast, cam, sc, gpu)
        R, color, radii, gb, bb, ib, depth = _C.rasterize_gaussians(
            rs.bg, ten["means3D"], e, ten["opacities"], ten["scales"], ten["rotations"], 1.0, e, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
            rs.tanfovy, H, W, ten["shs"], 3, rs.campos, False)
        return (R, color.clone(), depth.clone(), radii.clone()), _C.context_query("last_late")

    sizes = ((320, 240), (400, 304), (256, 192))
    _C.set_option("no_list_cut", 1)
    try:
        ref = {wh: render(*wh)[0] for wh in sizes}
    finally:
        _C.set_option("no_list_cut", 0)
    _C.set_option("list_cut_always", 1)
    try:
        for visit in range(3):
            for wh in sizes:
                out, late = render(*wh)
                assert out[0] == ref[wh][0] and all(torch.equal(a, b) for a, b in zip(out[1:], ref[wh][1:])), (visit, wh)
                if visit >= 1:
                    assert late > P // 8, (visit, wh, late)      # the size's table was kept while the other sizes were rendered
    finally:
        _C.set_option("list_cut_always", 0)


@pytest.mark.remembered_cut_only
def test_list_cut_with_cut_depths_but_no_late_gaussian(orc, scenes, rast, gpu):
    """a pose may hold cut depths while the scatter marks NO Gaussian late (rectangles of more than 64 tiles are
    never late; a Gaussian over a tile without a cut stays early).  The host then enqueues no second pass, so the blend must not
    cut any list short either: a scene of screen-filling Gaussians is rendered opaque (tiles saturate, cut depths are learned),
    then nearly transparent at the same pose -- every pixel looks far behind the remembered cut depth -- and must equal the oracle."""
    import torch
    from conftest import settings_from
    _C = rast._C
    P, NBIG, W, H = 33_000, 400, 320, 240             # (P >= the bucket depth sort's minimum: the cut rides on it)
    sc = scenes.synth(P, 931)
    rng = np.random.default_rng(932)
    sc["means3D"][NBIG:, :] = 50.0                     # everything but the big ones: far outside the frustum, culled
    sc["means3D"][:NBIG] = rng.uniform(-1.0, 1.0, size=(NBIG, 3)).astype(np.float32)
    sc["scales"][:NBIG] = rng.uniform(1.0, 1.5, size=(NBIG, 3)).astype(np.float32)     # sigma ~ 100 px: every rectangle is the whole image
    sc["opacities"][:NBIG] = 0.97
    cam = scenes.camera(1, 7, W, H)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)  # noqa: E731
    e = torch.empty(0)

    def render(scene):
        rs = settings_from(rast, cam, scene, gpu)
        ten = {k: t(scene[k]) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
        R, color, radii, gb, bb, ib, depth = _C.rasterize_gaussians(
            rs.bg, ten["means3D"], e, ten["opacities"], ten["scales"], ten["rotations"], 1.0, e, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
            rs.tanfovy, H, W, ten["shs"], 3, rs.campos, False)
        st = _C.debug_export(P, R, W, H, gb, bb, ib)
        return (R, color.clone(), depth.clone(), st["n_contrib"].clone(), st["final_T"].clone()), _C.context_query("last_late")

    _C.set_option("list_cut_always", 1)
    try:
        a, _ = render(sc)
        b, late_b = render(sc)                         # this one snapshots the cut depths the first left
        o = orc.render(sc, cam)
        assert (np.asarray(o["n_contrib"]) < 100).mean() > 0.9      # pixels stop a few dozen entries into their 400-entry lists: the tiles saturate and get cut depths
        assert late_b == 0                             # ... and still nobody is late
        assert np.array_equal(bits(b[1].cpu().numpy()), bits(o["out_color"]))
        fb0 = _C.context_query("cut_fallbacks")
        sc2 = dict(sc)
        sc2["opacities"] = (sc["opacities"] * 0.02).astype(np.float32)
        thin, late_c = render(sc2)
        o2 = orc.render(sc2, cam)
        assert late_c == 0 and _C.context_query("cut_fallbacks") == fb0
        assert thin[0] == o2["R"]
        assert np.array_equal(bits(thin[1].cpu().numpy()), bits(o2["out_color"]))
        assert np.array_equal(bits(thin[2].cpu().numpy()), bits(o2["out_depth"]))
        assert np.array_equal(bits(thin[4].cpu().numpy()), bits(np.asarray(o2["final_T"], dtype=np.float32)))     # (n_contrib counts positions of the CLIPPED lists: not comparable)
    finally:
        _C.set_option("list_cut_always", 0)


@pytest.mark.remembered_cut_only
def test_list_cut_under_a_changing_scene(orc, scenes, rast, gpu):
    """A training run changes the scene between two renders of a pose.  A fixed pose, twelve random edits in a row -- opacities scaled
    up or down, a tenth of the Gaussians pruned, the scene pushed away from / pulled towards the camera, a transparent and an opaque
    extreme -- each rendered under the list cut (whose cut depths come from the PREVIOUS edit's render) and with the cut switched off:
    outputs and per-pixel state bit for bit the same, whether the speculation held or the forward fell back; one edit is also taken
    through the backward against the oracle."""
    import torch
    from conftest import settings_from
    _C = rast._C
    P, W, H = 50_000, 256, 192
    base = scenes.synth(P, 905, scale_mul=1.2)
    cam = scenes.camera(5, 11, W, H)
    rng = np.random.default_rng(906)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)  # noqa: E731
    e = torch.empty(0)
    view_dir = np.asarray(cam["viewmatrix"], dtype=np.float32).reshape(4, 4)[:3, 2]     # (row-vector convention: column 2 = depth axis)

    def render(scene):
        rs = settings_from(rast, cam, scene, gpu)
        ten = {k: t(scene[k]) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
        R, color, radii, gb, bb, ib, depth = _C.rasterize_gaussians(
            rs.bg, ten["means3D"], e, ten["opacities"], ten["scales"], ten["rotations"], 1.0, e, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
            rs.tanfovy, H, W, ten["shs"], 3, rs.campos, False)
        st = _C.debug_export(P, R, W, H, gb, bb, ib)
        return R, color.clone(), depth.clone(), radii.clone(), st["n_contrib"].clone(), st["final_T"].clone()

    def same(a, b):
        return a[0] == b[0] and all(torch.equal(x, y) for x, y in zip(a[1:], b[1:]))

    _C.set_option("list_cut_always", 1)
    try:
        render(base)                                                 # the pose's first render: leaves cut depths
        fb0, cuts, sc = _C.context_query("cut_fallbacks"), 0, dict(base)
        for step in range(12):
            sc = dict(sc)
            kind = step % 6
            if kind == 0:
                sc["opacities"] = np.clip(sc["opacities"] * rng.uniform(0.3, 0.8), 0.0, 1.0).astype(np.float32)
            elif kind == 1:
                sc["opacities"] = np.clip(sc["opacities"] * rng.uniform(1.5, 3.0), 0.0, 1.0).astype(np.float32)
            elif kind == 2:
                op = sc["opacities"].copy(); op[rng.random(P) < 0.1] = 0.0; sc["opacities"] = op
            elif kind == 3:
                sc["means3D"] = (sc["means3D"] + view_dir * rng.uniform(-0.4, 0.4)).astype(np.float32)
            elif kind == 4:
                sc["opacities"] = (base["opacities"] * 0.02).astype(np.float32)       # nearly transparent: every tile looks far deeper
            else:
                sc["opacities"] = np.full_like(base["opacities"], 0.97)              # opaque: every tile saturates at once
            cut = render(sc)
            cuts += _C.context_query("last_late") > 0
            _C.set_option("no_list_cut", 1)
            try:
                ref = render(sc)
            finally:
                _C.set_option("no_list_cut", 0)
            assert same(cut, ref), f"edit {step} (kind {kind}): the list cut changed a result"
        assert cuts >= 8                                             # the cut really was in force most of the time ...
        assert _C.context_query("cut_fallbacks") > fb0               # ... and at least the transparent edits made it fall back
        g = scenes.upstream_grad(H, W, 907) * (H * W)
        o32 = orc.render(sc, cam, g)
        o64 = orc.render(sc, cam, g, f64=True)
        h = run_hip(rast, sc, cam, gpu, dL_dcolor=g, tile_clip=1)
        assert np.array_equal(bits(h["out_color"]), bits(o32["out_color"]))
        _check_grads(o64, o32, h, ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"])
    finally:
        _C.set_option("list_cut_always", 0)


def _poison_allocator(gpu, nbytes=256 << 20):
    """Fill the caching allocator's free blocks with NaN bit patterns, so that a torch.empty output that nobody writes is seen."""
    import torch
    t = [torch.full((nbytes // 16,), float("nan"), device=gpu) for _ in range(4)]
    small = [torch.full((n,), float("nan"), device=gpu) for n in (3000, 12000, 24000, 48000, 96000, 192000) for _ in range(4)]
    torch.cuda.synchronize()
    del t, small


@pytest.mark.parametrize("path", ["sh0", "sh3", "sh_M25", "colors_precomp", "cov3d_precomp"])
def test_backward_writes_every_row_of_poisoned_outputs(path, orc, scenes, rast, gpu):
    """The per-Gaussian backward does not read Gaussians whose gradient record is zero (options.dense_backward = 0, the default).
    Same gradients as the dense form and as the oracle; rows of culled / unlisted Gaussians exactly zero although the output
    arrays came out of the allocator full of NaNs."""
    P, W, H = 4000, 160, 120
    deg = {"sh0": 0, "sh3": 3, "sh_M25": 3}.get(path, 2)
    sc = scenes.synth(P, 77, sh_degree=deg)
    sc["means3D"][:500, 2] -= 40.0                       # behind the camera or far outside: culled
    sc["means3D"][500:900, 0] += 60.0
    kw = {}
    if path == "sh_M25":
        rng = np.random.default_rng(3)
        sc["shs"] = np.concatenate([sc["shs"], rng.normal(size=(P, 9, 3)).astype(np.float32)], 1)
    cam = scenes.camera(1, 4, W, H)
    if path == "colors_precomp":
        kw["colors_precomp"] = np.random.default_rng(4).uniform(0, 1, size=(P, 3)).astype(np.float32)
    if path == "cov3d_precomp":
        kw["cov3D_precomp"] = np.ascontiguousarray(orc.forward(sc, cam)["cov3D"]).astype(np.float32)
    g = scenes.upstream_grad(H, W, 78) * (H * W)
    o32 = orc.render(sc, cam, g, **kw)
    o64 = orc.render(sc, cam, g, f64=True, **kw)
    names = ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity"]
    names += ["dL_dcolors"] if path == "colors_precomp" else ["dL_dsh"]
    names += ["dL_dcov3D"] if path == "cov3d_precomp" else ["dL_dscales", "dL_drotations"]
    for dense in (0, 1):
        rast._C.set_option("dense_backward", dense)
        try:
            _poison_allocator(gpu)
            h = run_hip(rast, sc, cam, gpu, dL_dcolor=g, **kw)
        finally:
            rast._C.set_option("dense_backward", 0)
        for k in names:
            assert np.isfinite(h[k]).all(), (dense, k)
        _check_grads(o64, o32, h, names)
        dead = np.asarray(h["radii"]) <= 0
        assert dead.sum() >= 400
        for k in names:
            assert not np.asarray(h[k]).reshape(P, -1)[dead].any(), (dense, k)


@pytest.mark.parametrize("table", [0, 1], ids=["pose_table_off", "pose_table_on"])
@pytest.mark.parametrize("W,H", [(320, 240), (437, 251), (304, 528)])
def test_predicted_cut_is_verified_and_never_changes_a_result(table, W, H, orc, scenes, rast, gpu):
    """Round 5, option "tau_cut" (default on): a pose WITHOUT remembered cut depths -- one the context has never rendered, or any pose when
    the pose table is switched off -- gets PREDICTED cut depths from the call's own opacity mass per tile and coarse depth bin
    (gsrast_common.h: preprocess_fwd's histogram, tau_cut_kernel).  A prediction is a speculation like a remembered cut: the blend verifies it,
    a tile whose pixels do not all saturate in front of it is listed and blended again by the completion pass.  So: forwards under a predicted cut
    equal their cut-less twins bit for bit (state included) and the oracle; a scene built to fool the predictor -- dense on average, with an
    empty shaft along the view axis through the middle of the image, whose pixels never saturate -- is completed and equals the oracle too;
    gradients through a predicted-cut forward meet the oracle bar."""
    import torch
    from conftest import settings_from
    _C = rast._C
    P = 60_000
    sc = scenes.synth(P, 815, scale_mul=1.3)
    cam = scenes.camera(1, 7, W, H)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)  # noqa: E731
    e = torch.empty(0)

    def render(scene, camera=cam):
        rs = settings_from(rast, camera, scene, gpu)
        ten = {k: t(scene[k]) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
        R, color, radii, gb, bb, ib, depth = _C.rasterize_gaussians(
            rs.bg, ten["means3D"], e, ten["opacities"], ten["scales"], ten["rotations"], 1.0, e, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
            rs.tanfovy, H, W, ten["shs"], 3, rs.campos, False)
        st = _C.debug_export(P, R, W, H, gb, bb, ib)
        return (R, color.clone(), depth.clone(), radii.clone(), st["n_contrib"].clone(), st["final_T"].clone()), _C.context_query("last_late")

    def same(a, b):
        return a[0] == b[0] and all(torch.equal(x, y) for x, y in zip(a[1:], b[1:]))

    def twin(scene, camera=cam):
        _C.set_option("no_list_cut", 1)
        try:
            return render(scene, camera)[0]
        finally:
            _C.set_option("no_list_cut", 0)

    _C.set_option("list_cut_always", 1)
    _C.set_option("near_pose", 0)
    _C.set_option("no_order_hint", 0 if table else 1)
    try:
        for k in range(3):                            # the context learns its depth range and launch sizes on other poses of the ring
            render(sc, scenes.camera(3 + k, 7, W, H))
        full = twin(sc)
        o = orc.render(sc, cam)
        assert full[0] == o["R"] and np.array_equal(bits(full[1].cpu().numpy()), bits(o["out_color"]))
        p0 = _C.context_query("completion_passes")
        first, late_first = render(sc)                # never rendered by this context: PREDICTED cut depths
        assert late_first > P // 8, late_first       # the cube is opaque after a fraction of its depth, and the prediction sees that
        assert same(first, full)
        again, late_again = render(sc)                # (with the table on: the pose's remembered cut from now on)
        assert late_again > P // 8 and same(again, full)
        # an empty shaft along the view axis: the tiles in the middle of the image are dense ON AVERAGE, the pixels on the axis see nothing
        cpos = np.asarray(cam["campos"], dtype=np.float64)
        axis = -cpos / np.linalg.norm(cpos)
        rel = sc["means3D"].astype(np.float64) - cpos
        along = rel @ axis
        perp = np.linalg.norm(rel - np.outer(along, axis), axis=1)
        hole = dict(sc)
        hole["opacities"] = np.where((perp < 0.035 * along)[:, None], 1e-4, sc["opacities"]).astype(np.float32)
        oh = orc.render(hole, cam)
        got, late_h = render(hole, scenes.camera(1, 7, W, H))
        assert late_h > 0
        assert got[0] == oh["R"] and np.array_equal(bits(got[1].cpu().numpy()), bits(oh["out_color"])) and np.array_equal(bits(got[2].cpu().numpy()), bits(oh["out_depth"]))
        assert same(got, twin(hole))
        torch.cuda.synchronize()
        render(sc, scenes.camera(2, 7, W, H))         # (the report of a completion pass reaches the host with a later forward)
        print("completion passes:", _C.context_query("completion_passes") - p0, "tau_req", _C.context_query("tau_req"))
        # gradients through a forward under a predicted cut (another unseen pose), oracle bar
        cam_g = scenes.camera(6, 7, W, H)
        g = scenes.upstream_grad(H, W, 816) * (H * W)
        o32 = orc.render(sc, cam_g, g)
        o64 = orc.render(sc, cam_g, g, f64=True)
        h = run_hip(rast, sc, cam_g, gpu, dL_dcolor=g, tile_clip=1)
        assert np.array_equal(bits(h["out_color"]), bits(o32["out_color"]))
        _check_grads(o64, o32, h, ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"])
    finally:
        _C.set_option("list_cut_always", 0)
        _C.set_option("near_pose", 0)
        _C.set_option("no_order_hint", 0)


@pytest.mark.parametrize("P,W,H,deg", [(20000, 256, 192, 3), (5000, 97, 83, 1)])
def test_untouched_rows_are_zero_rows(P, W, H, deg, orc, scenes, rast, gpu):
    """Round 5: the forward blend keeps one bit per Gaussian, "no pixel consumed it" (GeomLayout::untouched); the backward writes those
    Gaussians' rows as zeros beside the blend backward and its per-Gaussian kernel works through the others only -- stateless, for every
    forward.  Forced on at test size (option late_fill_min_p = 0): every gradient at the oracle bar, untouched rows exactly zero, and the
    same gradients as with the bits switched off."""
    import torch
    _C = rast._C
    sc = scenes.synth(P, 820, sh_degree=deg, scale_mul=1.2)
    cam = scenes.camera(2, 5, W, H)
    g = scenes.upstream_grad(H, W, 821) * (H * W)
    o32 = orc.render(sc, cam, g)
    o64 = orc.render(sc, cam, g, f64=True)
    names = ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"]
    _C.set_option("late_fill_min_p", 0)
    try:
        h = run_hip(rast, sc, cam, gpu, dL_dcolor=g, tile_clip=1)
        _C.set_option("touch_bits", 0)
        try:
            h0 = run_hip(rast, sc, cam, gpu, dL_dcolor=g, tile_clip=1)
        finally:
            _C.set_option("touch_bits", 1)
    finally:
        _C.set_option("late_fill_min_p", 750000)
    assert np.array_equal(bits(h["out_color"]), bits(o32["out_color"]))
    _check_grads(o64, o32, h, names)
    for k in names:          # float-atomic order differs between two backwards: close, not bit-identical; zero rows are the same rows
        a, b = h[k].reshape(P, -1), h0[k].reshape(P, -1)
        assert np.array_equal((a != 0).any(1), (b != 0).any(1)), k
        assert (np.abs(a - b) <= 1e-6 + 1e-3 * np.abs(b)).all(), k


@pytest.mark.parametrize("late_fill", [True, False], ids=["grouped_backward", "per_gaussian_backward"])
@pytest.mark.parametrize("P,W,H,deg", [(20000, 256, 192, 3), (5000, 97, 83, 1)])
def test_stale_gradient_records_are_never_read(P, W, H, deg, late_fill, orc, scenes, rast, gpu):
    """Round 5: a forward that keeps untouched bits zeroes only the gradient records of the Gaussians some pixel consumed
    (grec_zero_touched_kernel); every other record holds whatever the buffer held before.  The state buffers are handed out filled with 0xFF
    bytes (NaN as floats): every gradient is finite, at the oracle bar, and the rows of untouched Gaussians are exactly zero -- through the
    grouped and the per-Gaussian form of the backward, and with the records zeroed whole (option sparse_grec = 0) for comparison."""
    _C = rast._C
    sc = scenes.synth(P, 830, sh_degree=deg, scale_mul=1.2)
    cam = scenes.camera(1, 5, W, H)
    g = scenes.upstream_grad(H, W, 831) * (H * W)
    o32 = orc.render(sc, cam, g)
    o64 = orc.render(sc, cam, g, f64=True)
    names = ["dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"]
    _C.set_option("late_fill_min_p", 0 if late_fill else 1 << 30)
    _C.POISON_STATE_BUFFERS = True
    try:
        assert _C.get_option("sparse_grec") == 1
        h = run_hip(rast, sc, cam, gpu, dL_dcolor=g, tile_clip=1)
        _C.set_option("sparse_grec", 0)
        try:
            h0 = run_hip(rast, sc, cam, gpu, dL_dcolor=g, tile_clip=1)
        finally:
            _C.set_option("sparse_grec", 1)
    finally:
        _C.POISON_STATE_BUFFERS = False
        _C.set_option("late_fill_min_p", 750000)
    assert np.array_equal(bits(h["out_color"]), bits(o32["out_color"]))
    for k in names:
        assert np.isfinite(h[k]).all(), k
    _check_grads(o64, o32, h, names)
    for k in names:
        a, b = h[k].reshape(P, -1), h0[k].reshape(P, -1)
        assert np.array_equal((a != 0).any(1), (b != 0).any(1)), k
        assert (np.abs(a - b) <= 1e-6 + 1e-3 * np.abs(b)).all(), k
