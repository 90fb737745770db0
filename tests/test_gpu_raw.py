"""-m gpu: the raw-parameter entry points (include/gsrast.h: gsrast_forward_raw / gsrast_backward_raw, module GaussianRasterizerRaw) --
SURVEY.md 8f rank 3 as written: the activation / deformation epilogue of scene/saro_gaussian.py:807-847 (activations :39-47) fused
into the per-Gaussian kernels, the [P,16,3] coefficient tensor never materialised.

Bars, in all 16 present / absent combinations of the four residuals:
  * rasterizer outputs BIT-IDENTICAL to fused_epilogue.activate_gaussians -> GaussianRasterizer (the unchanged drop-in path);
  * against  epilogue_oracle o gsrast_oracle : the forward bit-exact on the oracle's rendering of the activated attributes, every raw
    gradient within 1e-5 abs (+1e-4 relative) of  epilogue_oracle.backward( f64 rasterizer oracle )."""
import itertools

import numpy as np
import pytest
import torch

from conftest import settings_from
from gpu_harness import bits

pytestmark = pytest.mark.gpu

COMBOS = [dict(zip(("motion_res", "rot_res", "trbf", "shs_res"), c)) for c in itertools.product((True, False), repeat=4)]
from conftest import grad_tol


def _raw_scene(scenes, P, seed, M, deg):
    """Raw leaves whose activations give a scene like scenes.synth, plus small residuals."""
    sc = scenes.synth(P, seed, sh_degree=deg)
    rng = np.random.default_rng(seed + 1000)
    shs = sc["shs"][:, :M].astype(np.float32)
    raw = dict(
        xyz=sc["means3D"], motion_res=0.01 * rng.normal(size=(P, 3)),
        rotation=sc["rotations"] * rng.uniform(0.5, 2.0, size=(P, 1)), rot_res=np.concatenate([0.05 * rng.normal(size=(P, 4)), 0.1 * rng.normal(size=(P, 3))], 1),
        scaling=np.log(sc["scales"]), opacity=np.log(sc["opacities"].clip(1e-4, 1 - 1e-4) / (1 - sc["opacities"].clip(1e-4, 1 - 1e-4))),
        trbf=rng.uniform(0.3, 1.0, size=(P, 1)), f_dc=shs[:, :1], f_rest=shs[:, 1:], shs_res=0.03 * rng.normal(size=(P, M, 3)))
    raw["rotation"][0] = 0.0; raw["rot_res"][0, :4] = 0.0          # a zero quaternion: normalize clamps at eps
    return sc, {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in raw.items()}


def _tensors(raw, use, dev):
    t = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in raw.items()}
    kw = dict(motion_residual=t["motion_res"] if use["motion_res"] else None, rot_residual=t["rot_res"] if use["rot_res"] else None,
              trbfoutput=t["trbf"] if use["trbf"] else None, shs_residual=t["shs_res"] if use["shs_res"] else None)
    return t, kw


@pytest.mark.parametrize("use", COMBOS, ids=["".join("1" if v else "0" for v in c.values()) for c in COMBOS])
def test_raw_rasterizer_against_epilogue_then_rasterizer_and_the_oracles(use, orc, scenes, rast, gpu):
    from oracle import epilogue_oracle as eo
    import fused_epilogue
    P, W, H, M, deg = 3000, 160, 112, 16, 3
    sc, raw = _raw_scene(scenes, P, 401, M, deg)
    cam = scenes.camera(1, 4, W, H)
    rs = settings_from(rast, cam, sc, gpu)
    g_np = scenes.upstream_grad(H, W, 402)
    g = torch.from_numpy(g_np).to(gpu)

    # (a) the fused path -- half of the combinations with the default per-Gaussian backward (Gaussians with a zero gradient record
    # are not read), the other half with every Gaussian read
    ta, kwa = _tensors(raw, use, gpu)
    m2a = torch.zeros((P, 3), device=gpu, requires_grad=True)
    rast._C.set_option("dense_backward", 0 if (use["motion_res"] ^ use["shs_res"]) else 1)
    try:
        ca, ra, da = rast.GaussianRasterizerRaw(rs)(ta["xyz"], m2a, ta["rotation"], ta["scaling"], ta["opacity"], ta["f_dc"], ta["f_rest"], **kwa)
        ca.backward(g)
    finally:
        rast._C.set_option("dense_backward", 0)
    # (b) the drop-in path behind the standalone epilogue
    tb, kwb = _tensors(raw, use, gpu)
    m2b = torch.zeros((P, 3), device=gpu, requires_grad=True)
    motion, rot, scale, opa, shs = fused_epilogue.activate_gaussians(tb["xyz"], tb["rotation"], tb["scaling"], tb["opacity"], tb["f_dc"], tb["f_rest"], **kwb)
    cb, rb, db = rast.GaussianRasterizer(rs)(means3D=motion, means2D=m2b, opacities=opa, shs=shs, scales=scale, rotations=rot)
    cb.backward(g)
    torch.cuda.synchronize()
    assert torch.equal(ca, cb) and torch.equal(ra, rb) and torch.equal(da, db), "raw path must render bit-identically to epilogue -> rasterizer"
    assert int((ra > 0).sum()) > P // 2
    names = ["xyz", "rotation", "scaling", "opacity", "f_dc", "f_rest"] + [k for k in ("motion_res", "rot_res", "trbf", "shs_res") if use[k]]
    for k in names:
        a, b = ta[k].grad, tb[k].grad
        assert a is not None and a.shape == ta[k].shape, k
        sl = slice(1, None) if k in ("rotation", "rot_res") else slice(None)     # row 0: x / eps, 1e12-scaled
        assert ((a[sl] - b[sl]).abs() <= 2e-7 + 1e-4 * b[sl].abs()).all(), (k, float((a[sl] - b[sl]).abs().max()))
    for k in ("motion_res", "rot_res", "trbf", "shs_res"):
        if not use[k]:
            assert ta[k].grad is None
    assert ((m2a.grad - m2b.grad).abs() <= 2e-7 + 1e-4 * m2b.grad.abs()).all()

    # (c) epilogue_oracle o gsrast_oracle.  The rasterizer oracle renders the ACTIVATED attributes the device produced (fp32; the
    # standalone epilogue is itself checked against epilogue_oracle in test_epilogue.py) so that no 1-ulp input difference flips a
    # radius; the chain back to the raw leaves is epilogue_oracle.backward in fp64.
    act = dict(sc)
    act.update(means3D=motion.detach().cpu().numpy(), rotations=rot.detach().cpu().numpy(), scales=scale.detach().cpu().numpy(),
               opacities=opa.detach().cpu().numpy(), shs=shs.detach().cpu().numpy())
    eo_f = eo.forward(raw["xyz"], raw["rotation"], raw["scaling"], raw["opacity"], raw["f_dc"], raw["f_rest"],
                      motion_res=raw["motion_res"] if use["motion_res"] else None, rot_res=raw["rot_res"] if use["rot_res"] else None,
                      trbf=raw["trbf"] if use["trbf"] else None, shs_res=raw["shs_res"] if use["shs_res"] else None)
    np.testing.assert_allclose(act["scales"], eo_f["scale"], rtol=3e-6)
    np.testing.assert_allclose(act["shs"], eo_f["shs"], rtol=3e-6, atol=1e-7)
    o32 = orc.render(act, cam, g_np)
    o64 = orc.render(act, cam, g_np, f64=True)
    assert np.array_equal(ra.cpu().numpy(), o32["radii"])
    assert np.array_equal(bits(ca.detach().cpu().numpy()), bits(o32["out_color"]))
    assert np.array_equal(bits(da.detach().cpu().numpy()), bits(o32["out_depth"]))
    b64 = eo.backward(raw["rotation"], raw["scaling"], raw["opacity"], raw["rot_res"] if use["rot_res"] else None,
                      raw["trbf"] if use["trbf"] else None, o64["dL_drotations"], o64["dL_dscales"], o64["dL_dopacity"].reshape(P, 1))
    want = dict(xyz=o64["dL_dmeans3D"], rotation=b64["rotation"], scaling=b64["scaling"], opacity=b64["logit"].reshape(P, 1),
                f_dc=o64["dL_dsh"][:, :1], f_rest=o64["dL_dsh"][:, 1:])
    if use["motion_res"]:
        want["motion_res"] = o64["dL_dmeans3D"]
    if use["rot_res"]:
        want["rot_res"] = np.concatenate([b64["rotation"], b64["scaling"]], 1)
    if use["trbf"]:
        want["trbf"] = b64["trbf"].reshape(P, 1)
    if use["shs_res"]:
        want["shs_res"] = o64["dL_dsh"]
    # the same chain from the fp32 oracle's gradients: what an fp32 evaluation of the reference's formulas gives (the floor of conftest.grad_tol)
    b32 = eo.backward(raw["rotation"], raw["scaling"], raw["opacity"], raw["rot_res"] if use["rot_res"] else None,
                      raw["trbf"] if use["trbf"] else None, o32["dL_drotations"].astype(np.float64), o32["dL_dscales"].astype(np.float64),
                      o32["dL_dopacity"].astype(np.float64).reshape(P, 1))
    want32 = dict(xyz=o32["dL_dmeans3D"], rotation=b32["rotation"], scaling=b32["scaling"], opacity=b32["logit"].reshape(P, 1),
                  f_dc=o32["dL_dsh"][:, :1], f_rest=o32["dL_dsh"][:, 1:], motion_res=o32["dL_dmeans3D"],
                  rot_res=np.concatenate([b32["rotation"], b32["scaling"]], 1), trbf=b32["trbf"].reshape(P, 1) if use["trbf"] else None, shs_res=o32["dL_dsh"])
    for k, w in want.items():
        got = ta[k].grad.cpu().numpy().astype(np.float64).reshape(w.shape)
        sl = slice(1, None) if k in ("rotation", "rot_res") else slice(None)
        err = np.abs(got[sl] - w[sl])
        assert (err <= grad_tol(w[sl], np.asarray(want32[k], dtype=np.float64).reshape(w.shape)[sl])).all(), (k, float(err.max()))
    err = np.abs(m2a.grad.cpu().numpy() - o64["dL_dmeans2D"])
    assert (err <= grad_tol(o64["dL_dmeans2D"], o32["dL_dmeans2D"])).all()


@pytest.mark.parametrize("P,M,deg", [(1, 16, 3), (127, 16, 2), (129, 16, 3), (1001, 4, 1), (130, 4, 0)])
def test_raw_rasterizer_ragged_sizes_and_short_rows(P, M, deg, scenes, rast, gpu):
    """Blocks that end inside a 16-byte group of features_dc / features_rest (P not a multiple of 4), a single Gaussian, M = 4."""
    import fused_epilogue
    W, H = 96, 64
    sc, raw = _raw_scene(scenes, P, 411 + P, M, deg)
    raw["rotation"][0] = [1.0, 0.1, 0.0, 0.2]
    cam = scenes.camera(0, 3, W, H)
    rs = settings_from(rast, cam, sc, gpu)
    g = torch.from_numpy(scenes.upstream_grad(H, W, 412)).to(gpu)
    for use, dense in ((COMBOS[0], 0), (COMBOS[-1], 0), (COMBOS[0], 1)):
        ta, kwa = _tensors(raw, use, gpu)
        m2 = torch.zeros((P, 3), device=gpu, requires_grad=True)
        rast._C.set_option("dense_backward", dense)
        try:
            if not dense:       # the outputs come out of the allocator full of NaNs: rows of the Gaussians that are not read must still be zeros
                junk = [torch.full((n,), float("nan"), device=gpu) for n in (P, 3 * P, 4 * P, 7 * P, 45 * P, 48 * P) for _ in range(3)]
                del junk
            ca, ra, da = rast.GaussianRasterizerRaw(rs)(ta["xyz"], m2, ta["rotation"], ta["scaling"], ta["opacity"], ta["f_dc"], ta["f_rest"], **kwa)
            ca.backward(g)
        finally:
            rast._C.set_option("dense_backward", 0)
        tb, kwb = _tensors(raw, use, gpu)
        m2b = torch.zeros((P, 3), device=gpu, requires_grad=True)
        act = fused_epilogue.activate_gaussians(tb["xyz"], tb["rotation"], tb["scaling"], tb["opacity"], tb["f_dc"], tb["f_rest"], **kwb)
        cb, rb, db = rast.GaussianRasterizer(rs)(means3D=act[0], means2D=m2b, opacities=act[3], shs=act[4], scales=act[2], rotations=act[1])
        cb.backward(g)
        assert torch.equal(ca, cb) and torch.equal(ra, rb) and torch.equal(da, db)
        for k in ta:
            if ta[k].grad is None:
                assert tb[k].grad is None, k
                continue
            a, b = ta[k].grad, tb[k].grad
            assert ((a - b).abs() <= 2e-7 + 1e-4 * b.abs()).all(), (k, float((a - b).abs().max()))


def test_raw_entry_points_reject_bad_arguments(scenes, rast, gpu):
    P, W, H = 64, 48, 32
    sc, raw = _raw_scene(scenes, P, 421, 16, 3)
    cam = scenes.camera(0, 1, W, H)
    rs = settings_from(rast, cam, sc, gpu)
    t, _ = _tensors(raw, COMBOS[-1], gpu)
    m2 = torch.zeros((P, 3), device=gpu)
    R = rast.GaussianRasterizerRaw(rs)
    with pytest.raises(RuntimeError):       # M = 9 rows are not a multiple of 16 bytes: the fused path does not take them
        R(t["xyz"], m2, t["rotation"], t["scaling"], t["opacity"], t["f_dc"], t["f_rest"][:, :8].contiguous())
    with pytest.raises(RuntimeError):
        R(t["xyz"], m2, t["rotation"], t["scaling"], t["opacity"], t["f_dc"], t["f_rest"], rot_residual=torch.zeros((P, 4), device=gpu))
    with pytest.raises(RuntimeError):
        R(t["xyz"].cpu(), m2, t["rotation"], t["scaling"], t["opacity"], t["f_dc"], t["f_rest"])
    # evaluation: no gradient anywhere -> forward only, still the same picture
    with torch.no_grad():
        c0, r0, d0 = R(t["xyz"], m2, t["rotation"], t["scaling"], t["opacity"], t["f_dc"], t["f_rest"])
    c1, r1, d1 = R(t["xyz"], m2, t["rotation"], t["scaling"], t["opacity"], t["f_dc"], t["f_rest"])
    assert torch.equal(c0, c1.detach()) and torch.equal(r0, r1)


def test_raw_rasterizer_with_no_gaussians(scenes, rast, gpu):
    """P = 0 (rasterize_points.cu:81: nothing is rendered): zero image, empty radii, and a backward that returns empty gradients."""
    W, H = 64, 48
    sc = scenes.synth(4, 431)
    cam = scenes.camera(0, 1, W, H)
    rs = settings_from(rast, cam, sc, gpu)
    z = lambda *s: torch.zeros(s, device=gpu, requires_grad=True)  # noqa: E731
    xyz, rot, scl, opa, dc, rest = z(0, 3), z(0, 4), z(0, 3), z(0, 1), z(0, 1, 3), z(0, 15, 3)
    m2 = z(0, 3)
    c, r, d = rast.GaussianRasterizerRaw(rs)(xyz, m2, rot, scl, opa, dc, rest)
    assert c.shape == (3, H, W) and float(c.detach().abs().max()) == 0.0 and r.numel() == 0 and float(d.detach().abs().max()) == 0.0
    c.sum().backward()
    assert xyz.grad is not None and xyz.grad.shape == (0, 3)


@pytest.mark.remembered_cut_only
@pytest.mark.parametrize("use", [COMBOS[0], COMBOS[6], COMBOS[15]], ids=["1111", "1001", "0000"])
def test_raw_rasterizer_under_the_list_cut(use, scenes, rast, gpu):
    """The second and third forward of a pose through the raw entry points bin and COLOUR only the early Gaussians (include/gsrast.h:
    options.no_list_cut; the compacting colour kernel assembles cat(dc, rest) + residual for those alone): outputs bit-identical to the
    render without the cut, gradients of every raw leaf equal to its within fp32 accumulation order, also with every Gaussian read in the backward."""
    P, W, H, M, deg = 50_000, 256, 192, 16, 3
    sc, raw = _raw_scene(scenes, P, 451, M, deg)
    raw["opacity"] = raw["opacity"] + 2.0                      # denser: most tiles saturate
    cam = scenes.camera(2, 5, W, H)
    rs = settings_from(rast, cam, sc, gpu)
    g = torch.from_numpy(scenes.upstream_grad(H, W, 452) * (H * W)).to(gpu)
    names = ["xyz", "rotation", "scaling", "opacity", "f_dc", "f_rest"] + [k for k in ("motion_res", "rot_res", "trbf", "shs_res") if use[k]]

    def run(dense):
        t, kw = _tensors(raw, use, gpu)
        m2 = torch.zeros((P, 3), device=gpu, requires_grad=True)
        rast._C.set_option("dense_backward", dense)
        try:
            c, r, d = rast.GaussianRasterizerRaw(rs)(t["xyz"], m2, t["rotation"], t["scaling"], t["opacity"], t["f_dc"], t["f_rest"], **kw)
            late = rast._C.context_query("last_late")
            c.backward(g)
        finally:
            rast._C.set_option("dense_backward", 0)
        torch.cuda.synchronize()
        return c.detach(), r, d.detach(), {k: t[k].grad.clone() for k in names}, m2.grad.clone(), late

    rast._C.set_option("list_cut_always", 1)
    try:
        rast._C.set_option("no_list_cut", 1)                   # the reference: every Gaussian binned (leaves this pose's cut depths)
        try:
            first = run(0)
        finally:
            rast._C.set_option("no_list_cut", 0)
        assert first[5] == 0
        for dense in (0, 1):
            nxt = run(dense)
            assert nxt[5] > P // 4, nxt[5]
            assert torch.equal(nxt[0], first[0]) and torch.equal(nxt[1], first[1]) and torch.equal(nxt[2], first[2])
            for k in names:
                a, b = nxt[3][k].double(), first[3][k].double()
                assert torch.isfinite(a).all(), k
                assert ((a - b).abs() <= 1e-5 * max(1.0, float(b.abs().max())) + 1e-4 * b.abs()).all(), k
            assert ((nxt[4] - first[4]).abs() <= 1e-5 * max(1.0, float(first[4].abs().max())) + 1e-4 * first[4].abs()).all()
    finally:
        rast._C.set_option("list_cut_always", 0)
