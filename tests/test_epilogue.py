"""Fused activation / deformation epilogue (the "next" row before the rasterizer).

CPU: the numpy oracle against an independently written torch restatement of the reference's formulation
(scene/saro_gaussian.py:807-847, activations :39-47) and its autograd.  GPU (-m gpu): the HIP kernels through the
autograd wrapper against both, forward and backward, in every combination of present / absent residuals."""
import itertools

import numpy as np
import pytest
import torch
import torch.nn.functional as F


def _inputs(P, M, seed):
    rng = np.random.default_rng(seed)
    d = dict(
        xyz=rng.normal(size=(P, 3)), motion_res=0.05 * rng.normal(size=(P, 3)),
        rotation=rng.normal(size=(P, 4)), rot_res=0.1 * rng.normal(size=(P, 7)),
        scaling=rng.normal(-3.0, 1.0, size=(P, 3)), opacity=rng.normal(0.0, 2.0, size=(P, 1)),
        trbf=rng.uniform(0.0, 1.0, size=(P, 1)),
        f_dc=rng.uniform(-1.7, 1.7, size=(P, 1, 3)), f_rest=0.1 * rng.normal(size=(P, M - 1, 3)),
        shs_res=0.05 * rng.normal(size=(P, M, 3)))
    d["rotation"][0] = 0.0; d["rot_res"][0, :4] = 0.0        # a zero quaternion: normalize clamps at eps
    return {k: v.astype(np.float32) for k, v in d.items()}


def torch_epilogue(t, use):
    """The reference's own formulation, in torch (written for this test)."""
    motion = t["xyz"] + t["motion_res"] if use["motion_res"] else t["xyz"]
    if use["rot_res"]:
        rot = F.normalize(t["rotation"] + t["rot_res"][:, :4])
        scale = torch.exp(t["scaling"] + t["rot_res"][:, 4:])
    else:
        rot, scale = F.normalize(t["rotation"]), torch.exp(t["scaling"])
    opa = torch.sigmoid(t["opacity"]) * t["trbf"] if use["trbf"] else torch.sigmoid(t["opacity"])
    shs = torch.cat((t["f_dc"], t["f_rest"]), dim=1)
    if use["shs_res"]:
        shs = shs + t["shs_res"]
    return motion, rot, scale, opa, shs


COMBOS = [dict(zip(("motion_res", "rot_res", "trbf", "shs_res"), c)) for c in itertools.product((True, False), repeat=4)]


@pytest.mark.parametrize("use", [COMBOS[0], COMBOS[-1], COMBOS[5]], ids=["all", "none", "mixed"])
def test_numpy_oracle_matches_torch_restatement(use):
    from oracle import epilogue_oracle as eo
    P, M = 257, 16
    inp = _inputs(P, M, 5)
    t = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in inp.items()}
    outs = torch_epilogue(t, use)
    o = eo.forward(inp["xyz"], inp["rotation"], inp["scaling"], inp["opacity"], inp["f_dc"], inp["f_rest"],
                   motion_res=inp["motion_res"] if use["motion_res"] else None, rot_res=inp["rot_res"] if use["rot_res"] else None,
                   trbf=inp["trbf"] if use["trbf"] else None, shs_res=inp["shs_res"] if use["shs_res"] else None)
    for name, a in zip(("motion", "rot", "scale", "opacity", "shs"), outs):
        np.testing.assert_allclose(o[name][1:], a.detach().numpy()[1:], rtol=1e-13, atol=1e-15, err_msg=name)
    rng = np.random.default_rng(6)
    ups = [torch.from_numpy(rng.normal(size=tuple(a.shape))) for a in outs]
    torch.autograd.backward(outs, ups)
    b = eo.backward(inp["rotation"], inp["scaling"], inp["opacity"], inp["rot_res"] if use["rot_res"] else None,
                    inp["trbf"] if use["trbf"] else None, ups[1].numpy(), ups[2].numpy(), ups[3].numpy())
    np.testing.assert_allclose(b["rotation"][1:], t["rotation"].grad.numpy()[1:], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(b["scaling"], t["scaling"].grad.numpy(), rtol=1e-12)
    np.testing.assert_allclose(b["logit"], t["opacity"].grad.numpy(), rtol=1e-12, atol=1e-15)
    if use["trbf"]:
        np.testing.assert_allclose(b["trbf"], t["trbf"].grad.numpy(), rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("use", COMBOS, ids=["".join("1" if v else "0" for v in c.values()) for c in COMBOS])
@pytest.mark.parametrize("P,M", [(1000, 16), (333, 9)])
def test_hip_epilogue_matches_reference_formulation(use, P, M, gpu):
    from oracle import epilogue_oracle as eo
    import fused_epilogue
    inp = _inputs(P, M, 9)
    dev = gpu
    t = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in inp.items()}
    outs = fused_epilogue.activate_gaussians(
        t["xyz"], t["rotation"], t["scaling"], t["opacity"], t["f_dc"], t["f_rest"],
        motion_residual=t["motion_res"] if use["motion_res"] else None, rot_residual=t["rot_res"] if use["rot_res"] else None,
        trbfoutput=t["trbf"] if use["trbf"] else None, shs_residual=t["shs_res"] if use["shs_res"] else None)
    o = eo.forward(inp["xyz"], inp["rotation"], inp["scaling"], inp["opacity"], inp["f_dc"], inp["f_rest"],
                   motion_res=inp["motion_res"] if use["motion_res"] else None, rot_res=inp["rot_res"] if use["rot_res"] else None,
                   trbf=inp["trbf"] if use["trbf"] else None, shs_res=inp["shs_res"] if use["shs_res"] else None)
    for name, a in zip(("motion", "rot", "scale", "opacity", "shs"), outs):
        got = a.detach().cpu().numpy().astype(np.float64)
        sl = slice(1, None) if name == "rot" else slice(None)     # row 0: zero quaternion (0/eps), checked below
        np.testing.assert_allclose(got[sl], o[name][sl], rtol=3e-6, atol=1e-7, err_msg=name)
    assert not outs[1][0].any()                                    # normalize(0) = 0
    # backward against torch autograd of the reference's formulation on the same device (fp32) and the fp64 oracle
    rng = np.random.default_rng(10)
    ups = [torch.from_numpy(rng.normal(size=tuple(a.shape)).astype(np.float32)).to(dev) for a in outs]
    torch.autograd.backward(outs, ups)
    t2 = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in inp.items()}
    torch.autograd.backward(torch_epilogue(t2, use), ups)
    for k in t:
        used = use.get(k, True)
        if not used:
            assert t[k].grad is None
            continue
        a, b = t[k].grad, t2[k].grad
        sl = slice(1, None) if k in ("rotation", "rot_res") else slice(None)
        assert torch.allclose(a[sl], b[sl], rtol=2e-5, atol=1e-6), (k, float((a[sl] - b[sl]).abs().max()))
    b64 = eo.backward(inp["rotation"], inp["scaling"], inp["opacity"], inp["rot_res"] if use["rot_res"] else None,
                      inp["trbf"] if use["trbf"] else None, ups[1].cpu().numpy(), ups[2].cpu().numpy(), ups[3].cpu().numpy())
    np.testing.assert_allclose(t["scaling"].grad.cpu().numpy(), b64["scaling"], rtol=3e-6, atol=1e-7)
    np.testing.assert_allclose(t["opacity"].grad.cpu().numpy(), b64["logit"], rtol=3e-5, atol=1e-7)
    np.testing.assert_allclose(t["rotation"].grad.cpu().numpy()[1:], b64["rotation"][1:], rtol=3e-5, atol=1e-6)


@pytest.mark.gpu
def test_epilogue_feeds_the_rasterizer(scenes, rast, gpu):
    """End to end: raw parameters -> fused epilogue -> rasterizer -> backward reaches the raw parameters."""
    from conftest import settings_from
    import fused_epilogue
    P, W, H = 2000, 96, 64
    sc = scenes.synth(P, 121)
    cam = scenes.camera(0, 1, W, H)
    rs = settings_from(rast, cam, sc, gpu)
    tt = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)  # noqa: E731
    raw = dict(xyz=tt(sc["means3D"]), rotation=tt(sc["rotations"]) * 2.0, scaling=torch.log(tt(sc["scales"])),
               opacity=torch.logit(tt(sc["opacities"]).clamp(1e-4, 1 - 1e-4)), f_dc=tt(sc["shs"][:, :1]), f_rest=tt(sc["shs"][:, 1:]))
    raw = {k: v.requires_grad_(True) for k, v in raw.items()}
    motion, rot, scale, opa, shs = fused_epilogue.activate_gaussians(raw["xyz"], raw["rotation"], raw["scaling"], raw["opacity"],
                                                                      raw["f_dc"], raw["f_rest"])
    m2 = torch.zeros((P, 3), device=gpu, requires_grad=True)
    color, radii, depth = rast.GaussianRasterizer(rs)(means3D=motion, means2D=m2, opacities=opa, shs=shs, scales=scale, rotations=rot)
    color.sum().backward()
    for k, v in raw.items():
        assert v.grad is not None and torch.isfinite(v.grad).all() and v.grad.abs().sum() > 0, k
    assert torch.allclose(shs, tt(sc["shs"])) and torch.allclose(scale, tt(sc["scales"]), rtol=1e-5)
