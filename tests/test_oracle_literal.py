"""CPU: hand-computed scenes with LITERAL expected numbers for the edge semantics the reference's source fixes -- cheap extra pins of
oracle/gsrast_oracle.c on top of the reference-generated vectors (tests/test_oracle_golden.py).  Every expected value below was
worked out by hand from the cited lines of submodules/gaussian_rasterization_ch3/cuda_rasterizer/, not produced by the oracle.

Set-up shared by all scenes: identity view matrix, tan(fov/2) = 0.5, a 17 x 17 image (so focal = 17 / (2 * 0.5) = 17 by
rasterizer_impl.cu:222-223 and ndc 0 lands on pixel 8.0 exactly by auxiliary.h:41-44), isotropic Gaussians on the optical axis
(identity quaternion), colours given directly (colors_precomp).  Then for a Gaussian of scale s at depth z
(forward.cu:74-113): cov2D = (17 s / z)^2 + 0.3 on the diagonal, 0 off it; conic = 1 / that; at the centre pixel (8, 8) the
offset is 0, power = 0 and alpha = min(0.99, opacity) (forward.cu:338-347)."""
import math

import numpy as np
import pytest

W = H = 17
F32 = np.float32


def _cam(scenes):
    fov = 2.0 * math.atan(0.5)
    view = np.eye(4, dtype=np.float32)
    proj = scenes.projection(0.01, 100.0, fov, fov)
    return dict(image_height=H, image_width=W, tanfovx=0.5, tanfovy=0.5, viewmatrix=view,
                projmatrix=np.ascontiguousarray(view @ proj.T.astype(np.float32)), campos=np.zeros(3, np.float32),
                scale_modifier=1.0, prefiltered=False)


def _scene(zs, ss, ops, bg):
    n = len(zs)
    return dict(means3D=np.array([[0.0, 0.0, z] for z in zs], F32), scales=np.array([[s, s, s] for s in ss], F32),
                rotations=np.tile(np.array([1.0, 0.0, 0.0, 0.0], F32), (n, 1)), opacities=np.array(ops, F32).reshape(n, 1),
                shs=np.zeros((n, 16, 3), F32), sh_degree=0, bg=np.array(bg, F32))


def test_focal_projection_and_the_eigenvalue_floor(orc, scenes):
    """rasterizer_impl.cu:222-223 focal = size / (2 tan); forward.cu:229-232: lambda = mid + sqrt(max(0.1, mid^2 - det)) -- the 0.1
    floor is what makes this radius 6 (3 sqrt(2.5127 + sqrt(0.1)) = 5.05 -> 6) and not 5 (3 sqrt(2.5127) = 4.76)."""
    sc = _scene([2.0], [0.175], [0.8], (0, 0, 0))
    o = orc.render(sc, _cam(scenes), colors_precomp=np.array([[1.0, 0.0, 0.0]], F32))
    cov = (17.0 * 0.175 / 2.0) ** 2 + 0.3                       # 2.51265625
    assert abs(cov - 2.51265625) < 1e-12
    np.testing.assert_allclose(o["means2D"][0], [8.0, 8.0], atol=1e-5)
    np.testing.assert_allclose(o["conic_opacity"][0], [1 / cov, 0.0, 1 / cov, 0.8], rtol=2e-6, atol=1e-7)
    assert int(o["radii"][0]) == 6
    # getRect (auxiliary.h:46-56): min = int((8 - 6) / 16) = 0, max = int((8 + 6 + 15) / 16) = 1 in both axes: one tile of the 2 x 2 grid
    assert int(o["tiles_touched"][0]) == 1 and o["R"] == 1
    np.testing.assert_allclose(o["depths"][0], 2.0, atol=1e-6)
    np.testing.assert_allclose(o["cov3D"][0], [0.175 ** 2, 0, 0, 0.175 ** 2, 0, 0.175 ** 2], rtol=1e-6, atol=1e-9)


def test_transmittance_cutoff_leaves_last_contributor_alone(orc, scenes):
    """forward.cu:352-357: when T (1 - alpha) < 1e-4 the pixel is done BEFORE last_contributor is updated, so the Gaussian that
    triggered the cut-off is counted in `contributor` but not in n_contrib, and the backward (backward.cu:486-488) never visits it.
    Three opaque Gaussians: alpha = 0.99 each (clamp).  1 - 0.99f = 0.009999990463256836; the second would leave
    T = 0.00999999^2 = 9.99998e-5 < 1e-4 -> cut.  n_contrib = 1, final_T = 0.00999999, colour = 0.99 c1 + T bg."""
    sc = _scene([2.0, 3.0, 4.0], [0.2, 0.2, 0.2], [1.0, 1.0, 1.0], (0.5, 0.5, 0.5))
    cols = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1]], F32)
    g = np.zeros((3, H, W), F32)
    g[:, 8, 8] = 1.0
    cam = _cam(scenes)
    for f64 in (False, True):
        o = orc.render(sc, cam, g, colors_precomp=cols, f64=f64)
        T1 = float(F32(1.0) - F32(0.99))
        assert int(o["n_contrib"].reshape(H, W)[8, 8]) == 1
        np.testing.assert_allclose(o["final_T"].reshape(H, W)[8, 8], T1, rtol=1e-6)
        np.testing.assert_allclose(o["out_color"][:, 8, 8], [0.99 + 0.5 * T1, 0.5 * T1, 0.5 * T1], rtol=1e-6)
        # median depth (forward.cu:368-372): T = 1 > 0.5 and T (1 - alpha) < 0.5 at the first Gaussian -> its depth
        np.testing.assert_allclose(o["out_depth"][0, 8, 8], 2.0, rtol=1e-6)
        # backward, centre pixel only, dL/dpixel = (1, 1, 1):
        #   dL/dalpha = (c1 - 0) . dpix * T_before(= 1)  -  T_final / (1 - alpha) * (bg . dpix)  =  1 - 1.5  =  -0.5   (backward.cu:519-534)
        #   dL/dopacity += G dL/dalpha with G = exp(0) = 1 and NO mask for the 0.99 clamp (backward.cu:554)      -> -0.5
        #   dL/dcolour = alpha T dpix = 0.99 (backward.cu:506-513)
        np.testing.assert_allclose(o["dL_dopacity"][0, 0], -0.5, rtol=1e-5)
        np.testing.assert_allclose(o["dL_dcolors"][0], [0.99, 0.99, 0.99], rtol=1e-6)
        # the second Gaussian triggered the cut-off, the third was never reached: neither receives anything
        for k in ("dL_dopacity", "dL_dcolors", "dL_dmeans2D", "dL_dconic", "dL_dmeans3D", "dL_dscales", "dL_drotations"):
            assert not np.any(o[k][1:]), k
        # power = 0 at the pixel centre: d = 0, so the position / shape gradients of this pixel vanish (backward.cu:540-551)
        assert not np.any(o["dL_dmeans2D"][0]) and not np.any(o["dL_dconic"][0])


def test_median_depth_default_and_crossing(orc, scenes):
    """forward.cu:308 D = 15.0 until the transmittance crosses 0.5 (forward.cu:368-372).  Centre pixel, opacities 0.3 then 0.4:
    T = 1 -> 0.7 -> 0.42, the crossing happens at the SECOND Gaussian (depth 3).  A single 0.3 Gaussian never crosses: 15.0.
    The corner pixel (0, 0) is 8 px away in x and y: power = -0.5 * 128 / 3.19 = -20.06, alpha = 0.3 e^-20 < 1/255: skipped
    (forward.cu:344-347), nothing contributes: colour = bg, final_T = 1, n_contrib = 0, depth 15.0."""
    cam = _cam(scenes)
    cols = np.array([[0.2, 0.4, 0.6], [0.9, 0.1, 0.3]], F32)
    o = orc.render(_scene([2.0, 3.0], [0.2, 0.2], [0.3, 0.4], (0.1, 0.2, 0.3)), cam, colors_precomp=cols)
    np.testing.assert_allclose(o["out_depth"][0, 8, 8], 3.0, rtol=1e-6)
    np.testing.assert_allclose(o["final_T"].reshape(H, W)[8, 8], 0.7 * 0.6, rtol=1e-6)
    want = 0.3 * cols[0] + 0.7 * 0.4 * cols[1] + 0.42 * np.array([0.1, 0.2, 0.3])
    np.testing.assert_allclose(o["out_color"][:, 8, 8], want, rtol=2e-6)
    assert int(o["n_contrib"].reshape(H, W)[8, 8]) == 2
    assert float(o["out_depth"][0, 0, 0]) == 15.0 and int(o["n_contrib"].reshape(H, W)[0, 0]) == 0
    assert float(o["final_T"].reshape(H, W)[0, 0]) == 1.0
    np.testing.assert_allclose(o["out_color"][:, 0, 0], [0.1, 0.2, 0.3], rtol=1e-6)
    o1 = orc.render(_scene([2.0], [0.2], [0.3], (0, 0, 0)), cam, colors_precomp=cols[:1])
    assert float(o1["out_depth"][0, 8, 8]) == 15.0
    np.testing.assert_allclose(o1["final_T"].reshape(H, W)[8, 8], 0.7, rtol=1e-6)


def test_conic_gradient_slots_and_off_centre_pixel(orc, scenes):
    """backward.cu:549-551 writes dL/dconic into .x, .y and .w of the float4; .z is never written.  One Gaussian (s = 0.2, z = 2:
    conic a = c = 1 / 3.19), dL/dpixel = (1, 0, 0) at pixel (10, 8) only: d = (8 - 10, 8 - 8) = (-2, 0),
    power = -0.5 a 4 = -2 a, G = exp(-2 a), alpha = 0.5 G, dL/dalpha = c_r T(= 1) = 1 (bg = 0), dL/dG = 0.5,
    gdx = G d.x = -2 G:   dL/dconic.x = -0.5 gdx d.x dL/dG = -0.5 (-2 G)(-2)(0.5) = -G;  .y = -0.5 gdx d.y dL/dG = 0;  .w = 0.
    dL/dmean2D.x = dL/dG * (-gdx a - gdy b) * 0.5 W = 0.5 * 2 G a * 8.5."""
    cam = _cam(scenes)
    g = np.zeros((3, H, W), F32)
    g[0, 8, 10] = 1.0
    a = 1.0 / ((17 * 0.2 / 2.0) ** 2 + 0.3)
    G = math.exp(-2.0 * a)
    o = orc.render(_scene([2.0], [0.2], [0.5], (0, 0, 0)), cam, g, colors_precomp=np.array([[1.0, 0.5, 0.25]], F32), f64=True)
    dc = o["dL_dconic"][0]
    np.testing.assert_allclose(dc[0], -G, rtol=1e-5)
    assert dc[2] == 0.0 and abs(dc[1]) < 1e-12 and abs(dc[3]) < 1e-12
    np.testing.assert_allclose(o["dL_dmeans2D"][0, 0], 0.5 * (2.0 * G * a) * 8.5, rtol=1e-5)
    assert abs(o["dL_dmeans2D"][0, 1]) < 1e-12 and o["dL_dmeans2D"][0, 2] == 0.0
    np.testing.assert_allclose(o["dL_dopacity"][0, 0], G, rtol=1e-5)       # G dL/dalpha
