"""Helpers for the -m gpu parity tests: run the HIP path through the drop-in API / C ABI and
collect every intermediate array in the reference's layout (numpy) next to the oracle's."""
import numpy as np
import torch

from conftest import settings_from


def run_hip(rast, scene, cam, device, dL_dcolor=None, colors_precomp=None, cov3D_precomp=None, exp_mode=0,
            tile_clip=0):
    """tile_clip=0: the reference's literal tile lists (every tile of the 3-sigma square), so keys / point_list /
    ranges / n_contrib can be compared entry by entry.  tile_clip=1 is the product default (lists without the tiles the
    alpha >= 1/255 ellipse cannot reach): same outputs, shorter lists -- see check_clipped_lists."""
    _C = rast._C
    _C.set_option("exp_mode", exp_mode)
    _C.set_option("tile_clip", tile_clip)
    _C.set_option("debug_state", 1)         # the forward also keeps cov3D for debug_export (the product default does not store it)
    try:
        return _run_hip(rast, scene, cam, device, dL_dcolor, colors_precomp, cov3D_precomp)
    finally:
        _C.set_option("tile_clip", 1)
        _C.set_option("debug_state", 0)


def _run_hip(rast, scene, cam, device, dL_dcolor, colors_precomp, cov3D_precomp):
    _C = rast._C
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)  # noqa: E731
    rs = settings_from(rast, cam, scene, device)
    P = scene["means3D"].shape[0]
    leaf = {}

    def mk(name, arr):
        x = t(arr).requires_grad_(dL_dcolor is not None)
        leaf[name] = x
        return x

    means3D = mk("means3D", scene["means3D"])
    means2D = torch.zeros((P, 3), dtype=torch.float32, device=device, requires_grad=dL_dcolor is not None)
    leaf["means2D"] = means2D
    opac = mk("opacities", scene["opacities"])
    kw = {}
    if colors_precomp is None:
        kw["shs"] = mk("shs", scene["shs"])
    else:
        kw["colors_precomp"] = mk("colors_precomp", colors_precomp)
    if cov3D_precomp is None:
        kw["scales"] = mk("scales", scene["scales"])
        kw["rotations"] = mk("rotations", scene["rotations"])
    else:
        kw["cov3D_precomp"] = mk("cov3D_precomp", cov3D_precomp)
    r = rast.GaussianRasterizer(raster_settings=rs)

    # call the autograd function directly too, to get at the state buffers
    empty = torch.empty(0)
    args = (rs.bg, means3D.detach(), kw.get("colors_precomp", empty).detach(), opac.detach(),
            kw.get("scales", empty).detach(), kw.get("rotations", empty).detach(), rs.scale_modifier,
            kw.get("cov3D_precomp", empty).detach(), rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy,
            rs.image_height, rs.image_width, kw.get("shs", empty).detach(), rs.sh_degree, rs.campos, False)
    # (the state that is exported comes from a forward WITHOUT the list cut: under a cut -- remembered or, round 5, predicted -- the lists hold
    # the early Gaussians only and a completed tile's range lies in the point list's second half; the module call below may run under one,
    # and must produce the same image bit for bit)
    _C.set_option("no_list_cut", 1)
    try:
        R, color0, radii0, gb, bb, ib, depth0 = _C.rasterize_gaussians(*args)
    finally:
        _C.set_option("no_list_cut", 0)
    out = {"R": R}
    if P > 0:
        st = _C.debug_export(P, R, rs.image_width, rs.image_height, gb, bb, ib)
        out.update({k: v.cpu().numpy() for k, v in st.items()})
        out["keys_sorted"] = out["keys_sorted"].view(np.uint64)
        out["point_list"] = out["point_list"].view(np.uint32)
        out["ranges"] = out["ranges"].view(np.uint32)
        out["n_contrib"] = out["n_contrib"].view(np.uint32)
        out["tiles_touched"] = out["tiles_touched"].view(np.uint32)

    color, radii, depth = r(means3D=means3D, means2D=means2D, opacities=opac, **kw)
    torch.cuda.synchronize()
    assert torch.equal(color, color0) and torch.equal(radii, radii0) and torch.equal(depth, depth0), \
        "two identical forward calls must be bit-identical"
    out.update(out_color=color.detach().cpu().numpy(), radii=radii.cpu().numpy(), out_depth=depth.detach().cpu().numpy())
    if dL_dcolor is not None:
        color.backward(t(dL_dcolor))
        torch.cuda.synchronize()
        names = dict(means3D="dL_dmeans3D", means2D="dL_dmeans2D", opacities="dL_dopacity", shs="dL_dsh",
                     colors_precomp="dL_dcolors", scales="dL_dscales", rotations="dL_drotations",
                     cov3D_precomp="dL_dcov3D")
        for k, x in leaf.items():
            out[names[k]] = x.grad.detach().cpu().numpy() if x.grad is not None else None
    return out


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def check_clipped_lists(o, h, W, H):
    """Lists built with tile_clip=1 (h) against the reference's literal lists (o, oracle):
    every tile's list is an ordered subsequence of the reference's, and every pixel's last contributor
    (the entry n_contrib points at) is the same Gaussian."""
    gx = (W + 15) // 16
    ro, rh = o["ranges"].reshape(-1, 2).astype(np.int64), h["ranges"].reshape(-1, 2).astype(np.int64)
    po, ph = o["point_list"], h["point_list"]
    total = int((rh[:, 1] - rh[:, 0]).sum())
    assert total <= o["R"]
    for t in range(ro.shape[0]):
        lo, lh = po[ro[t, 0]:ro[t, 1]], ph[rh[t, 0]:rh[t, 1]]
        assert len(lh) <= len(lo)
        if len(lh) == 0:
            continue
        # ordered subsequence: positions of lh's entries in lo must be strictly increasing (ids are unique per tile)
        pos = {int(g): i for i, g in enumerate(lo)}
        idx = np.array([pos.get(int(g), -1) for g in lh])
        assert (idx >= 0).all() and (np.diff(idx) > 0).all(), f"tile {t}: clipped list is not a subsequence"
    nco, nch = o["n_contrib"].reshape(H, W).astype(np.int64), h["n_contrib"].reshape(H, W).astype(np.int64)
    ys, xs = np.mgrid[0:H, 0:W]
    tile = (ys // 16) * gx + xs // 16
    assert ((nco > 0) == (nch > 0)).all()
    m = nco > 0
    last_o = po[(ro[tile, 0] + nco - 1)[m]]
    last_h = ph[(rh[tile, 0] + nch - 1)[m]]
    np.testing.assert_array_equal(last_o, last_h)
    return total
