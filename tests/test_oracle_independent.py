"""CPU: the oracle against an INDEPENDENT derivation (tests/math_renderer.py: torch float64, autograd, no tile lists, no
hand-written gradients -- written from the mathematics, sharing no code with oracle/gsrast_oracle.c or the HIP kernels).

What this pins that the reference-derived golden vectors cannot (no CUDA here): computeCov2D / conic / radius / getRect
(forward.cu:74-113, :218-236, auxiliary.h:46-56), the compositing recurrence (forward.cu:311-381) and EVERY backward formula
(backward.cu:144-341, :399-557) -- the latter as "the oracle's hand-derived gradients equal autograd's".  A misreading of the
reference shared by the oracle and the kernels would have to be shared by the textbook formulas too to stay green."""
import numpy as np
import pytest
import torch

import math_renderer as mr


def _t(a):
    return torch.as_tensor(np.asarray(a, np.float64))


def _scene(scenes, P, seed, W, H, k, V, deg=3, scale_mul=1.0, bg=(0.0, 0.0, 0.0), opac_mul=1.0):
    sc = scenes.synth(P, seed, sh_degree=deg, scale_mul=scale_mul)
    sc["bg"] = np.array(bg, np.float32)
    sc["opacities"] = (sc["opacities"] * opac_mul).astype(np.float32)
    return sc, scenes.camera(k, V, W, H)


def _unclamped(sc, cam):
    """Gaussians whose centre is inside 1.3 x the field of view: outside, the reference's backward knowingly drops
    d(clamp * t.z)/dt.z (forward.cu:82-87 vs backward.cu:175-176), which autograd keeps."""
    V = cam["viewmatrix"].astype(np.float64)
    tv = sc["means3D"].astype(np.float64) @ V[:3, :3] + V[3, :3]
    return (np.abs(tv[:, 0] / tv[:, 2]) < 1.29 * cam["tanfovx"]) & (np.abs(tv[:, 1] / tv[:, 2]) < 1.29 * cam["tanfovy"])


@pytest.mark.parametrize("P,seed,W,H,k,V,smul", [(400, 11, 96, 64, 1, 5, 1.0), (900, 12, 80, 112, 2, 7, 0.6), (120, 13, 64, 48, 0, 3, 2.5)])
def test_projection_conic_radius_rect_against_the_math(orc, scenes, P, seed, W, H, k, V, smul):
    sc, cam = _scene(scenes, P, seed, W, H, k, V, scale_mul=smul)
    o32 = orc.forward(sc, cam)
    o64 = orc.forward(sc, cam, st32=o32)
    pr = mr.project(_t(sc["means3D"]), _t(sc["scales"]), _t(sc["rotations"]), cam)
    d = pr["disc"]
    # discrete quantities: equal unless the real-valued radius / rectangle edge sits within fp32 rounding of an integer
    firm = (d["radius_margin"] > 1e-4) & (d["rect_margin"] > 1e-5)
    assert firm.mean() > 0.98
    np.testing.assert_array_equal(o32["radii"][firm], d["radius"][firm])
    np.testing.assert_array_equal(o32["tiles_touched"][firm].astype(np.int64), d["tiles"][firm])
    vis = (o32["radii"] > 0) & firm
    assert vis.sum() > P // 4
    # continuous quantities, fp64 oracle build vs fp64 math: rounding only
    np.testing.assert_allclose(o64["means2D"][vis], pr["pix"].numpy()[vis], rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(o64["depths"][vis], pr["depth"].numpy()[vis], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(o64["conic_opacity"][vis, :3], pr["conic"].numpy()[vis], rtol=1e-9, atol=1e-12)
    S = pr["Sigma"].numpy()
    cov6 = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], axis=1)
    np.testing.assert_allclose(o64["cov3D"][vis], cov6[vis], rtol=1e-12, atol=1e-15)
    # ... and the fp32 build (what the HIP kernel is bit-compared with) within fp32 rounding of the math
    np.testing.assert_allclose(o32["conic_opacity"][vis, :3], pr["conic"].numpy()[vis], rtol=3e-4, atol=1e-6)
    assert (np.abs(o32["cov3D"][vis] - cov6[vis]) <= 1e-6 * np.abs(cov6[vis]).max(axis=1, keepdims=True)).all()   # off-diagonals cancel


CASES = [
    dict(P=300, seed=21, W=64, H=48, k=1, V=5, deg=3, smul=0.8, bg=(0.1, 0.2, 0.3), omul=1.0),
    dict(P=700, seed=22, W=96, H=80, k=3, V=7, deg=2, smul=0.5, bg=(0.0, 0.0, 0.0), omul=0.7),
    dict(P=150, seed=23, W=48, H=64, k=0, V=4, deg=1, smul=1.6, bg=(1.0, 1.0, 1.0), omul=1.0),
    dict(P=500, seed=24, W=80, H=64, k=2, V=6, deg=0, smul=0.7, bg=(0.0, 0.5, 0.0), omul=0.5),
]


@pytest.mark.parametrize("c", CASES, ids=lambda c: f"P{c['P']}_deg{c['deg']}")
def test_forward_and_all_gradients_against_autograd_of_the_math(orc, scenes, c):
    sc, cam = _scene(scenes, c["P"], c["seed"], c["W"], c["H"], c["k"], c["V"], c["deg"], c["smul"], c["bg"], c["omul"])
    W, H, P = c["W"], c["H"], c["P"]
    leaves = {n: _t(sc[n]).clone().requires_grad_(True) for n in ("means3D", "scales", "rotations", "opacities", "shs")}
    off = torch.zeros((P, 2), dtype=torch.float64, requires_grad=True)
    out = mr.render(leaves["means3D"], leaves["scales"], leaves["rotations"], leaves["opacities"], leaves["shs"], c["deg"], cam,
                    sc["bg"], ndc_offset=off)
    amb = out["ambiguous"]
    assert amb.mean() < 0.02, "too many pixels with an fp32-ambiguous decision for a meaningful comparison"
    assert out["min_depth_gap"] > 2e-6, "two Gaussians closer in depth than fp32 resolves: pick another seed"
    g = (scenes.upstream_grad(H, W, c["seed"] + 1) * (H * W)).astype(np.float32)
    g[:, amb] = 0.0                                     # ambiguous pixels take no part in the gradient, on either side
    o64 = orc.render(sc, cam, g, f64=True)
    # same Gaussians drawn, same order inside every tile
    o32 = orc.forward(sc, cam)
    d = out["proj"]["disc"]
    assert np.array_equal(o32["radii"] > 0, d["vis"]) and np.array_equal(o32["radii"], d["radius"]), "radius decision differs: pick another seed"
    keep = ~amb
    np.testing.assert_allclose(o64["out_color"][:, keep], out["color"].detach().numpy()[:, keep], rtol=0, atol=1e-10)
    np.testing.assert_allclose(o64["final_T"][keep], out["final_T"].detach().numpy()[keep], rtol=0, atol=1e-10)
    np.testing.assert_allclose(o64["out_depth"][0][keep], out["depth"].numpy()[keep], rtol=0, atol=1e-9)
    # backward: autograd of the math vs the oracle's hand-derived formulas
    (out["color"] * _t(g)).sum().backward()
    unc = _unclamped(sc, cam)
    pairs = {"means3D": "dL_dmeans3D", "scales": "dL_dscales", "rotations": "dL_drotations", "opacities": "dL_dopacity", "shs": "dL_dsh"}
    for name, key in pairs.items():
        got = o64[key].reshape(leaves[name].shape)
        want = leaves[name].grad.numpy()
        sel = unc if name == "means3D" else np.ones(P, bool)
        scale = max(1.0, float(np.abs(want[sel]).max()))
        np.testing.assert_allclose(got[sel], want[sel], rtol=1e-7, atol=1e-9 * scale, err_msg=name)
    np.testing.assert_allclose(o64["dL_dmeans2D"][:, :2], off.grad.numpy(), rtol=1e-7, atol=1e-9 * max(1.0, float(np.abs(off.grad.numpy()).max())))
    assert np.abs(leaves["means3D"].grad.numpy()[unc]).max() > 1e-3      # the comparison is not vacuous
    assert out["n_live"].max() >= 5


def test_clamp_passthrough_matters_only_where_alpha_saturates(orc, scenes):
    """backward.cu:538 / :554 apply no mask for alpha clamped at 0.99: with saturating Gaussians the oracle follows the reference
    (straight-through), and the two conventions differ exactly when such pairs exist."""
    sc, cam = _scene(scenes, 60, 31, 48, 48, 1, 4, deg=0, scale_mul=3.0)
    sc["opacities"][:] = 0.999
    t = {n: _t(sc[n]).clone().requires_grad_(True) for n in ("means3D", "scales", "rotations", "opacities", "shs")}
    out = mr.render(t["means3D"], t["scales"], t["rotations"], t["opacities"], t["shs"], 0, cam, sc["bg"])
    assert out["clamped_pairs"] > 0
    g = (scenes.upstream_grad(48, 48, 32) * (48 * 48)).astype(np.float32)
    g[:, out["ambiguous"]] = 0.0
    (out["color"] * _t(g)).sum().backward()
    o64 = orc.render(sc, cam, g, f64=True)
    want = t["opacities"].grad.numpy()
    np.testing.assert_allclose(o64["dL_dopacity"], want, rtol=1e-7, atol=1e-9 * max(1.0, float(np.abs(want).max())))
