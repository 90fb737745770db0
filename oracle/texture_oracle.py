"""TEST INFRASTRUCTURE ONLY (oracle): numpy fp64 restatement of the mip-mapped plane lookup the reference's residual
field performs through `nvdiffrast.torch.texture` (last item).  Only tests/, __graft_entry__.smoke()
and bench.py's baseline legs may import this file; the product path (saro-gs_amd/fused_hexplane.py over
csrc/gsrast_hexplane.h) never does.

PARITY UNPINNED against the dependency itself: nvdiffrast is neither vendored in /root/reference nor installed in this
image, and the reference pins no version of it (README.md:29 only links its install page), so its binary cannot be run here.
What is restated is its PUBLISHED algorithm (NVlabs/nvdiffrast, texture op, `filter_mode='linear-mipmap-linear'`,
`boundary_mode='clamp'`, mip level from `mip_level_bias` only), anchored on the reference's own call site:

  scene/hexplane.py:26-60   grid_sample_wrapper: tex = grid[1,C,H,W] -> [1,H,W,C]; uv = coords; bias = min over the
                            plane's two coordinates of `levels`; boundary_mode="clamp"; max_mip_level = 7 (spatial
                            plane) or 0 (plane with a time axis); filter_mode left at 'auto' = linear-mipmap-linear
                            because a bias is given
  scene/hexplane.py:95-139  interpolate_ms_features: six planes (itertools.combinations(range(4), 2)) summed per scale,
                            scales concatenated
  scene/hexplane.py:237-249 get_level: level = log2(2 * clamp(scale) / base_scale), 0 for the time axis

Published algorithm, as restated here:
  * mip stack: level l+1 = 2x2 box average of level l (2x1 / 1x2 when one extent is already 1); built while either
    extent is > 1 and l < max_mip_level; an odd extent > 1 cannot be halved (the op rejects it: error here as well)
  * level: flevel = clamp(bias, 0, n_levels); level0 = floor(flevel); if flevel > 0: level1 = min(level0 + 1, n_levels),
    f = flevel - level0; else single level
  * per level: texel space u = uv.x * w - 0.5, clamped to [0, w - 1]; i0 = floor(u), i1 = i0 + 1 unless the clamp hit
    (then i1 = i0: zero uv gradient), fractional weights; bilinear; out = a + f * (b - a)
  * gradients: to the texels of both levels (then pulled down the stack to level 0, the transpose of the box average),
    to uv (bilinear slope * extent, 0 where clamped), to the bias (sum_c dy_c (b_c - a_c) where f > 0)

The structure (box-average pyramid, border-clamped half-texel-centred bilinear, level lerp) is pinned against torch's own
`avg_pool2d` + `grid_sample(padding_mode='border', align_corners=False)` + autograd in tests/test_oracle_texture.py.
"""
import itertools

import numpy as np


def mip_sizes(W, H, max_mip_level):
    """[(w, h)] for levels 0..n_levels."""
    sizes = [(int(W), int(H))]
    w, h = int(W), int(H)
    while (w > 1 or h > 1) and len(sizes) - 1 < max_mip_level:
        if (w > 1 and (w & 1)) or (h > 1 and (h & 1)):
            raise ValueError(f"mip level {len(sizes) - 1} has an odd extent ({w}x{h}): limit max_mip_level")
        w, h = max(w >> 1, 1), max(h >> 1, 1)
        sizes.append((w, h))
    return sizes


def build_mips(tex, max_mip_level):
    """tex [H, W, C] -> list of levels (fp64)."""
    tex = np.asarray(tex, np.float64)
    H, W, _ = tex.shape
    out = [tex]
    for (w, h) in mip_sizes(W, H, max_mip_level)[1:]:
        p = out[-1]
        ph, pw = p.shape[:2]
        if pw > 1 and ph > 1:
            n = 0.25 * (p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2])
        elif pw > 1:
            n = 0.5 * (p[:, 0::2] + p[:, 1::2])
        else:
            n = 0.5 * (p[0::2] + p[1::2])
        assert n.shape[:2] == (h, w)
        out.append(n)
    return out


def pull_down(dmips):
    """Transpose of build_mips: gradient of all levels -> gradient of level 0."""
    d = [np.array(x, np.float64) for x in dmips]
    for l in range(len(d) - 1, 0, -1):
        g, p = d[l], d[l - 1]
        ph, pw = p.shape[:2]
        if pw > 1 and ph > 1:
            for dy, dx in itertools.product((0, 1), (0, 1)):
                p[dy::2, dx::2] += 0.25 * g
        elif pw > 1:
            p[:, 0::2] += 0.5 * g
            p[:, 1::2] += 0.5 * g
        else:
            p[0::2] += 0.5 * g
            p[1::2] += 0.5 * g
    return d[0]


def _level_terms(uv, w, h):
    u = uv[:, 0] * w - 0.5
    v = uv[:, 1] * h - 0.5
    u = np.minimum(np.maximum(u, 0.0), w - 1.0)
    v = np.minimum(np.maximum(v, 0.0), h - 1.0)
    cu = (u == 0.0) | (u == w - 1.0)
    cv = (v == 0.0) | (v == h - 1.0)
    iu0 = np.floor(u).astype(np.int64)
    iv0 = np.floor(v).astype(np.int64)
    iu1 = iu0 + np.where(cu, 0, 1)
    iv1 = iv0 + np.where(cv, 0, 1)
    return iu0, iu1, iv0, iv1, u - iu0, v - iv0


def _levels(bias, n_levels):
    fl = np.minimum(np.maximum(np.asarray(bias, np.float64), 0.0), float(n_levels))
    l0 = np.floor(fl).astype(np.int64)
    two = fl > 0.0
    l1 = np.where(two, np.minimum(l0 + 1, n_levels), 0)
    f = np.where(two, fl - l0, 0.0)
    return l0, l1, f


def texture(tex, uv, bias, max_mip_level, dy=None):
    """tex [H,W,C], uv [N,2], bias [N] -> out [N,C]; with dy [N,C] also (dtex [H,W,C], duv [N,2], dbias [N])."""
    mips = build_mips(tex, max_mip_level)
    uv = np.asarray(uv, np.float64)
    N, C = uv.shape[0], mips[0].shape[2]
    l0, l1, f = _levels(bias, len(mips) - 1)
    out = np.zeros((N, C))
    want = dy is not None
    if want:
        dy = np.asarray(dy, np.float64)
        dm = [np.zeros_like(m) for m in mips]
        duv = np.zeros((N, 2))
        dbias = np.zeros(N)
    vals = {}
    for which, lv in ((0, l0), (1, l1)):
        B = np.zeros((N, C))
        slope = np.zeros((N, 2))
        for l in np.unique(lv):
            sel = np.nonzero(lv == l)[0] if which == 0 else np.nonzero((lv == l) & (f > 0.0))[0]
            if sel.size == 0:
                continue
            m = mips[l]
            h, w = m.shape[:2]
            iu0, iu1, iv0, iv1, fu, fv = _level_terms(uv[sel], w, h)
            a00, a10, a01, a11 = m[iv0, iu0], m[iv0, iu1], m[iv1, iu0], m[iv1, iu1]
            fu_, fv_ = fu[:, None], fv[:, None]
            top = a00 + (a10 - a00) * fu_
            bot = a01 + (a11 - a01) * fu_
            B[sel] = top + (bot - top) * fv_
            if want:
                wl = (1.0 - f[sel]) if which == 0 else f[sel]
                g = dy[sel] * wl[:, None]
                np.add.at(dm[l], (iv0, iu0), g * (1 - fu_) * (1 - fv_))
                np.add.at(dm[l], (iv0, iu1), g * fu_ * (1 - fv_))
                np.add.at(dm[l], (iv1, iu0), g * (1 - fu_) * fv_)
                np.add.at(dm[l], (iv1, iu1), g * fu_ * fv_)
                dBdu = ((a10 - a00) * (1 - fv_) + (a11 - a01) * fv_) * w
                dBdv = ((a01 - a00) * (1 - fu_) + (a11 - a10) * fu_) * h
                duv[sel, 0] += (g * dBdu).sum(1)
                duv[sel, 1] += (g * dBdv).sum(1)
        vals[which] = B
    two = (f > 0.0)[:, None]
    out = np.where(two, vals[0] + f[:, None] * (vals[1] - vals[0]), vals[0])
    if not want:
        return out
    dbias = np.where(f > 0.0, (dy * (vals[1] - vals[0])).sum(1), 0.0)
    return out, pull_down(dm), duv, dbias


PLANES = list(itertools.combinations(range(4), 2))      # scene/hexplane.py:105-107: (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)


def interpolate_ms_features(pts, ms_grids, levels, dy=None):
    """scene/hexplane.py:95-139 with concat_features=True, concat_planes=False (the values ScaleAwareResField fixes,
    :168-169).  pts [N,4] in texture coordinates, ms_grids[scale][plane] = [C, H, W] (the parameter's layout without its
    leading 1), levels [N,4].  Returns features [N, sum C]; with dy also the per-grid gradients (same nesting)."""
    pts = np.asarray(pts, np.float64)
    levels = np.asarray(levels, np.float64)
    feats, dgrids = [], []
    off = 0
    for grids in ms_grids:
        C = grids[0].shape[0]
        acc = np.zeros((pts.shape[0], C))
        dg = []
        for ci, comb in enumerate(PLANES):
            tex = np.transpose(np.asarray(grids[ci], np.float64), (1, 2, 0))          # hexplane.py:35 permute(0,2,3,1)
            bias = levels[:, list(comb)].min(axis=1)                                    # hexplane.py:46
            mm = 7 if 3 not in comb else 0                                              # hexplane.py:55, :117
            if dy is None:
                acc = acc + texture(tex, pts[:, list(comb)], bias, mm)
            else:
                o, dt, _, _ = texture(tex, pts[:, list(comb)], bias, mm, dy[:, off:off + C])
                acc = acc + o
                dg.append(np.transpose(dt, (2, 0, 1)))
        feats.append(acc)
        dgrids.append(dg)
        off += C
    f = np.concatenate(feats, axis=1)
    return f if dy is None else (f, dgrids)
