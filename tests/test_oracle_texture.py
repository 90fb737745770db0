"""CPU: pins oracle/texture_oracle.py's structure (box-average mip stack, border-clamped bilinear with half-texel
centres, level lerp, and all three gradients) against torch's own avg_pool2d + grid_sample + autograd in fp64.
(The dependency it restates, nvdiffrast, is absent from the image: see the oracle's header.)"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import texture_oracle as tor


def torch_texture(tex, uv, bias, max_mip_level):
    """tex [H,W,C] (requires_grad ok), uv [N,2], bias [N]: pyramid by avg_pool2d, border bilinear per level, lerp."""
    t = tex.permute(2, 0, 1)[None]                     # [1,C,H,W]
    mips = [t]
    while (mips[-1].shape[2] > 1 or mips[-1].shape[3] > 1) and len(mips) - 1 < max_mip_level:
        h, w = mips[-1].shape[2:]
        mips.append(F.avg_pool2d(mips[-1], (2 if h > 1 else 1, 2 if w > 1 else 1)))
    n = len(mips) - 1
    fl = bias.clamp(0.0, float(n))
    l0 = fl.detach().floor().long()
    l1 = torch.clamp(l0 + 1, max=n)
    f = fl - l0
    grid = (2.0 * uv - 1.0)[None, None]                # [1,1,N,2]
    per_level = [F.grid_sample(m, grid, mode="bilinear", padding_mode="border", align_corners=False)[0, :, 0].t() for m in mips]
    stack = torch.stack(per_level)                     # [L,N,C]
    idx = torch.arange(uv.shape[0])
    a, b = stack[l0, idx], stack[l1, idx]
    return a + f[:, None] * (b - a)


@pytest.mark.parametrize("W,H,C,mm", [(16, 16, 5, 7), (32, 8, 3, 7), (8, 32, 4, 2), (12, 10, 3, 0), (64, 64, 4, 3), (4, 1, 2, 7)])
def test_oracle_matches_torch_autograd(W, H, C, mm):
    rng = np.random.default_rng(W * 131 + H)
    N = 400
    tex = rng.normal(size=(H, W, C))
    uv = rng.uniform(-0.1, 1.1, size=(N, 2))           # some points beyond the border (clamp)
    uv[:5] = [[0.0, 0.0], [1.0, 1.0], [0.5 / W, 0.5 / H], [1 - 0.5 / W, 0.3], [0.3, 1.0]]
    n_levels = len(tor.mip_sizes(W, H, mm)) - 1
    bias = rng.uniform(-0.5, n_levels + 0.7, size=N)
    bias[5:9] = [0.0, float(n_levels), 0.25, n_levels - 0.25]
    dy = rng.normal(size=(N, C))
    out, dtex, duv, dbias = tor.texture(tex, uv, bias, mm, dy)
    t = torch.tensor(tex, requires_grad=True)
    u = torch.tensor(uv, requires_grad=True)
    b = torch.tensor(bias, requires_grad=True)
    o = torch_texture(t, u, b, mm)
    o.backward(torch.tensor(dy))
    assert np.abs(out - o.detach().numpy()).max() < 1e-12
    assert np.abs(dtex - t.grad.numpy()).max() < 1e-11
    # torch's border mode lets the uv gradient through at an exactly-clamped coordinate on the inside; the restated op zeroes
    # it there (i1 = i0).  Compare away from exact clamps.
    inside = np.ones(N, bool)
    inside[:5] = False
    assert np.abs(duv - u.grad.numpy())[inside].max() < 1e-9
    # bias gradient: torch's clamp passes gradient at the closed ends, the op's `flevel > 0` test does not; and an exactly
    # integral level has f = 0 (single level).  Compare strictly inside.
    strict = (bias > 0) & (bias < n_levels) & (bias != np.floor(bias))
    if n_levels:
        assert np.abs(dbias - b.grad.numpy())[strict].max() < 1e-10
    assert np.all(dbias[~((bias > 0) & (bias < n_levels))] == 0.0)


def test_mip_rules():
    assert tor.mip_sizes(64, 64, 7) == [(64 >> l, 64 >> l) for l in range(7)]
    assert tor.mip_sizes(512, 512, 7)[-1] == (4, 4)
    assert tor.mip_sizes(64, 150, 0) == [(64, 150)]
    assert tor.mip_sizes(8, 2, 7) == [(8, 2), (4, 1), (2, 1), (1, 1)]
    with pytest.raises(ValueError):
        tor.mip_sizes(12, 10, 7)                       # 6x5: odd extent cannot be halved
    rng = np.random.default_rng(0)
    d = [rng.normal(size=(8 >> l, 8 >> l, 2)) for l in range(4)]
    tex = rng.normal(size=(8, 8, 2))
    # pull_down is the transpose of build_mips: <build(tex), d> == <tex, pull_down(d)> (+ level 0 identity)
    lhs = sum((m * g).sum() for m, g in zip(tor.build_mips(tex, 3), d))
    assert abs(lhs - (tex * tor.pull_down(d)).sum()) < 1e-12


def test_field_layout():
    """interpolate_ms_features: planes summed per scale, scales concatenated, time planes unmipped (hexplane.py:95-139)."""
    rng = np.random.default_rng(3)
    reso = [8, 8, 8, 6]
    grids = [[rng.normal(size=(3, reso[b] * m, reso[a] * m if a < 3 else reso[a])) if b < 3 else rng.normal(size=(3, reso[b], reso[a] * m))
              for (a, b) in tor.PLANES] for m in (1, 2)]
    N = 50
    pts = rng.uniform(0, 1, size=(N, 4))
    levels = np.concatenate([rng.uniform(0, 3, size=(N, 3)), np.zeros((N, 1))], axis=1)
    f = tor.interpolate_ms_features(pts, grids, levels)
    assert f.shape == (N, 6)
    want = sum(tor.texture(np.transpose(grids[1][ci], (1, 2, 0)), pts[:, list(c)], levels[:, list(c)].min(1), 7 if 3 not in c else 0)
               for ci, c in enumerate(tor.PLANES))
    assert np.abs(f[:, 3:] - want).max() < 1e-12
    dy = rng.normal(size=(N, 6))
    f2, dg = tor.interpolate_ms_features(pts, grids, levels, dy)
    assert np.abs(f2 - f).max() == 0 and dg[1][2].shape == grids[1][2].shape
    # gradient check by linearity: <dy, F(G + E) - F(G)> == <dG, E>
    E = [[rng.normal(size=g.shape) for g in gs] for gs in grids]
    f3 = tor.interpolate_ms_features(pts, [[g + e for g, e in zip(gs, es)] for gs, es in zip(grids, E)], levels)
    lhs = (dy * (f3 - f)).sum()
    rhs = sum((d * e).sum() for ds, es in zip(dg, E) for d, e in zip(ds, es))
    assert abs(lhs - rhs) < 1e-9 * max(1.0, abs(lhs))
