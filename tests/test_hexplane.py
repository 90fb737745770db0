"""Mip-mapped feature-plane lookup of the residual field (first item): the HIP kernels through
saro-gs_amd/fused_hexplane.py against oracle/texture_oracle.py (numpy fp64 restatement of the published texture op, pinned
against torch in tests/test_oracle_texture.py).  Tolerances: values 1e-5 abs; gradients 1e-5 of the largest entry (texel
gradients are sums over hundreds of points)."""
import numpy as np
import pytest
import torch

from oracle import texture_oracle as tor


def _points(rng, N, n_levels):
    uv = rng.uniform(-0.05, 1.05, size=(N, 2)).astype(np.float32)
    uv[:6] = [[0, 0], [1, 1], [0.5, 0.5], [1.0, 0.25], [0.25, 0.0], [0.999999, 0.000001]]
    lv = rng.uniform(-0.5, n_levels + 0.7, size=(N, 2)).astype(np.float32)
    lv[6:10] = [[0, 3], [n_levels, n_levels], [0.25, 0.5], [n_levels - 0.25, n_levels + 3]]
    return uv, lv


def _close(got, want, tol, what):
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(np.asarray(got, np.float64) - want).max())
    assert err <= tol * scale, f"{what}: max err {err:.3e} > {tol:.0e} * {scale:.3g}"


def _plane_case(gpu, W, H, C, mm, N, seed, scatter):
    import fused_hexplane
    from diff_gaussian_rasterization_ch3 import _C
    rng = np.random.default_rng(seed)
    n_levels = len(tor.mip_sizes(W, H, mm)) - 1
    tex = rng.normal(size=(C, H, W)).astype(np.float32)
    uv, lv = _points(rng, N, n_levels)
    dy = rng.normal(size=(N, C)).astype(np.float32)
    bias = lv.min(axis=1)
    out, dtex, duv, dbias = tor.texture(np.transpose(tex, (1, 2, 0)), uv, bias, mm, dy)
    assert _C.lib().gsrast_set_option(b"hexplane_scatter", int(scatter)) == 0
    try:
        g = torch.tensor(tex[None], device=gpu, requires_grad=True)
        p = torch.tensor(uv, device=gpu, requires_grad=True)
        l = torch.tensor(lv, device=gpu, requires_grad=True)
        o = fused_hexplane.texture_planes(p, l, [g], [(0, 1)], [mm], [0], C)
        o.backward(torch.tensor(dy, device=gpu))
        torch.cuda.synchronize()
    finally:
        _C.lib().gsrast_set_option(b"hexplane_scatter", 0)
    _close(o.detach().cpu().numpy(), out, 1e-5, "features")
    _close(g.grad[0].cpu().numpy(), np.transpose(dtex, (2, 0, 1)), 1e-5, "dL/dtex")
    # uv / bias gradients: compare where the oracle's discrete choices (floor of the texel coordinate and of the level) are
    # the same in fp32 -- away from texel centres and integral levels by more than fp32 resolution
    safe = np.ones(N, bool)
    for l_ in range(n_levels + 1):
        w, h = tor.mip_sizes(W, H, mm)[l_]
        for k, e in ((0, w), (1, h)):
            x = uv[:, k].astype(np.float64) * e - 0.5
            safe &= (np.abs(x - np.round(x)) > 1e-3) | (x < -1e-3) | (x > e - 1 + 1e-3)
    safe &= np.abs(bias - np.round(bias)) > 1e-4
    _close(p.grad.cpu().numpy()[safe], duv[safe], 2e-5, "dL/duv")
    dl = l.grad.cpu().numpy()
    first = lv[:, 0] <= lv[:, 1]
    got_bias = np.where(first, dl[:, 0], dl[:, 1])
    other = np.where(first, dl[:, 1], dl[:, 0])
    _close(got_bias[safe], dbias[safe], 2e-5, "dL/dbias")
    assert np.all(other == 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,C,mm", [(64, 64, 32, 7), (32, 8, 16, 7), (16, 25, 32, 0), (8, 2, 4, 7), (128, 128, 8, 7), (64, 128, 64, 0),
                                      (256, 256, 32, 7), (1, 1, 4, 7)])
@pytest.mark.parametrize("scatter", [0, 1])
def test_single_plane_against_oracle(gpu, W, H, C, mm, scatter):
    """scatter 0 = sorted runs (default), 1 = direct global atomics."""
    _plane_case(gpu, W, H, C, mm, 3001, W * 7 + H + C + mm, scatter)


@pytest.mark.gpu
def test_large_plane(gpu):
    _plane_case(gpu, 512, 512, 32, 7, 20000, 5, 0)
    _plane_case(gpu, 512, 512, 32, 7, 20000, 6, 1)


def _field(rng, reso, C, mults):
    return [[rng.normal(size=(1, C, (reso[b] * m if b < 3 else reso[b]), (reso[a] * m if a < 3 else reso[a]))).astype(np.float32)
             for (a, b) in tor.PLANES] for m in mults]


@pytest.mark.gpu
@pytest.mark.parametrize("reso,C,mults", [([64, 64, 64, 25], 32, (1,)), ([16, 16, 16, 10], 16, (1, 2, 4)), ([64, 64, 64, 128], 32, (1, 2))])
def test_field_against_oracle(gpu, reso, C, mults):
    """interpolate_ms_features (hexplane.py:95-139): 6 planes x scales in one launch, and dL/dgrid of every plane."""
    import fused_hexplane
    rng = np.random.default_rng(len(mults) * 100 + C)
    grids = _field(rng, reso, C, mults)
    N = 5000
    pts = rng.uniform(0, 1, size=(N, 4)).astype(np.float32)
    levels = np.concatenate([rng.uniform(0, np.log2(reso[0]), size=(N, 3)), np.zeros((N, 1))], axis=1).astype(np.float32)   # get_level :237-249
    dy = rng.normal(size=(N, C * len(mults))).astype(np.float32)
    want, dg = tor.interpolate_ms_features(pts, [[g[0] for g in gs] for gs in grids], levels, dy)
    params = [[torch.tensor(g, device=gpu, requires_grad=True) for g in gs] for gs in grids]
    out = fused_hexplane.interpolate_ms_features(torch.tensor(pts, device=gpu), params, 2, True, torch.tensor(levels, device=gpu), None)
    assert tuple(out.shape) == (N, C * len(mults))
    out.backward(torch.tensor(dy, device=gpu))
    _close(out.detach().cpu().numpy(), want, 1e-5, "features")
    for s, gs in enumerate(params):
        for ci, g in enumerate(gs):
            _close(g.grad[0].cpu().numpy(), dg[s][ci], 1e-5, f"dL/dgrid scale {s} plane {ci}")


@pytest.mark.gpu
def test_reference_call_shapes_and_layouts(gpu):
    """grid_sample_wrapper mirror (hexplane.py:26-60); channels_last parameters are used in place and receive their gradient;
    summed scales / concat_planes variants of interpolate_ms_features."""
    import fused_hexplane
    rng = np.random.default_rng(11)
    C, N = 8, 700
    tex = rng.normal(size=(1, C, 32, 16)).astype(np.float32)
    uv = rng.uniform(0, 1, size=(N, 2)).astype(np.float32)
    lv = rng.uniform(0, 4, size=(N, 2)).astype(np.float32)
    want = tor.texture(np.transpose(tex[0], (1, 2, 0)), uv, lv.min(1), 7)
    g = torch.tensor(tex, device=gpu).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    o = fused_hexplane.grid_sample_wrapper(g, torch.tensor(uv, device=gpu), torch.tensor(lv, device=gpu), True)
    assert tuple(o.shape) == (N, C)
    _close(o.detach().cpu().numpy(), want, 1e-5, "wrapper")
    o.sum().backward()
    assert g.grad.shape == g.shape
    _, dtex, _, _ = tor.texture(np.transpose(tex[0], (1, 2, 0)), uv, lv.min(1), 7, np.ones((N, C)))
    _close(g.grad[0].cpu().numpy(), np.transpose(dtex, (2, 0, 1)), 1e-5, "channels_last grad")
    o0 = fused_hexplane.grid_sample_wrapper(g, torch.tensor(uv, device=gpu), torch.tensor(lv, device=gpu), False)
    _close(o0.detach().cpu().numpy(), tor.texture(np.transpose(tex[0], (1, 2, 0)), uv, lv.min(1), 0), 1e-5, "time-plane wrapper")
    # summed scales (concat_features=False) and space|time blocks (concat_planes=True)
    grids = _field(rng, [8, 8, 8, 6], 4, (1, 2))
    pts = rng.uniform(0, 1, size=(N, 4)).astype(np.float32)
    levels = np.concatenate([rng.uniform(0, 3, size=(N, 3)), np.zeros((N, 1))], axis=1).astype(np.float32)
    params = [[torch.tensor(x, device=gpu) for x in gs] for gs in grids]
    per = [[tor.texture(np.transpose(gs[ci][0], (1, 2, 0)), pts[:, list(c)], levels[:, list(c)].min(1), 7 if 3 not in c else 0)
            for ci, c in enumerate(tor.PLANES)] for gs in grids]
    tp, tl = torch.tensor(pts, device=gpu), torch.tensor(levels, device=gpu)
    summed = fused_hexplane.interpolate_ms_features(tp, params, 2, False, tl, None)
    _close(summed.cpu().numpy(), sum(sum(p) for p in per), 1e-5, "summed scales")
    blocks = fused_hexplane.interpolate_ms_features(tp, params, 2, True, tl, None, concat_planes=True)
    want_b = np.concatenate([np.concatenate([p[0] + p[1] + p[3], p[2] + p[4] + p[5]], axis=1) for p in per], axis=1)
    _close(blocks.cpu().numpy(), want_b, 1e-5, "concat_planes")
    one = fused_hexplane.interpolate_ms_features(tp, params, 2, True, tl, 1)
    _close(one.cpu().numpy(), sum(per[0]), 1e-5, "num_levels=1")


@pytest.mark.gpu
def test_errors_and_empty(gpu):
    import fused_hexplane
    g = torch.zeros((1, 4, 10, 12), device=gpu)
    uv = torch.rand((5, 2), device=gpu)
    with pytest.raises(RuntimeError, match="odd extent"):
        fused_hexplane.texture_planes(uv, uv, [g], [(0, 1)], [7], [0], 4)       # 6x5 cannot be halved
    fused_hexplane.texture_planes(uv, uv, [g], [(0, 1)], [1], [0], 4)            # one level is fine
    with pytest.raises(RuntimeError, match="power of two"):
        fused_hexplane.texture_planes(uv, uv, [torch.zeros((1, 12, 8, 8), device=gpu)], [(0, 1)], [0], [0], 12)
    with pytest.raises(RuntimeError, match="GPU"):
        fused_hexplane.texture_planes(uv.cpu(), uv.cpu(), [g.cpu()], [(0, 1)], [0], [0], 4)
    e = torch.zeros((0, 2), device=gpu)
    gg = torch.ones((1, 4, 8, 8), device=gpu, requires_grad=True)
    o = fused_hexplane.texture_planes(e, e, [gg], [(0, 1)], [7], [0], 4)
    assert tuple(o.shape) == (0, 4)
    o.sum().backward()
    assert float(gg.grad.abs().max()) == 0.0
    # non-finite coordinates stay inside the plane (clamped), no fault
    bad = torch.tensor([[float("nan"), float("inf")], [-float("inf"), 0.5]], device=gpu)
    o = fused_hexplane.texture_planes(bad, torch.zeros_like(bad), [gg], [(0, 1)], [7], [0], 4)
    torch.cuda.synchronize()
    assert torch.isfinite(o).all()
