"""Mip-mapped feature-plane lookup of the scale-aware residual field on the HIP library -- the un-vendored
`nvdiffrast.torch.texture` call of the reference and the plane loop around it (first item).

Mirrors /root/reference/scene/hexplane.py:
  grid_sample_wrapper(grid, coords, levels, spatio_only)                      :26-60   one plane
  interpolate_ms_features(pts, ms_grids, grid_dimensions, concat_features,
                          levels, num_levels, concat_planes=False)            :95-139  all planes of all scales
with the same names, argument meaning and results; `interpolate_ms_features` runs the whole field as ONE forward launch
(after the per-level mip builds) instead of one texture op + add per plane and a cat per scale.  Gradients flow to the
plane parameters, and to `pts` / `levels` when they require them (the reference detaches both, saro_gaussian.py:780).

Planes are read channel-last ([1,H,W,C], what hexplane.py:35 builds with permute + contiguous on every call).  Parameters
created or converted with `memory_format=torch.channels_last` are used in place; others are copied like the reference does.
There is no CPU / PyTorch fallback: without libgsrast_hip.so and a GPU tensor this raises."""
import ctypes as C
import itertools
from typing import List, Optional, Sequence

import torch

from diff_gaussian_rasterization_ch3 import _C as _lib


def _channel_last(grid: torch.Tensor) -> torch.Tensor:
    """[1,C,H,W] (any strides) -> contiguous [H,W,C] view / copy (hexplane.py:29-35)."""
    if grid.dim() == 3:
        grid = grid.unsqueeze(0)
    if grid.dim() != 4 or grid.shape[0] != 1:
        raise RuntimeError("plane must be [1, C, H, W]")
    return grid.permute(0, 2, 3, 1).contiguous()[0]


class _PlaneSet:
    """Host-side descriptor array for gsrast_hexplane_* (include/gsrast.h: gsrast_plane)."""

    def __init__(self, texs: Sequence[torch.Tensor], cols: Sequence[Sequence[int]], max_mips: Sequence[int], offsets: Sequence[int],
                 grads: Optional[Sequence[torch.Tensor]] = None):
        self.n = len(texs)
        self.arr = (_lib.PlaneStruct * self.n)()
        for i, t in enumerate(texs):
            H, W, _ = t.shape
            self.arr[i] = _lib.PlaneStruct(t.data_ptr(), grads[i].data_ptr() if grads is not None else None, W, H,
                                           int(cols[i][0]), int(cols[i][1]), int(max_mips[i]), int(offsets[i]))


class _HexplaneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, levels, cols, max_mips, offsets, F, *grids):
        dev = pts.device
        if not pts.is_cuda:
            raise RuntimeError("hexplane lookup: tensors must be on a GPU (HIP) device; there is no CPU fallback")
        L = _lib.lib()
        texs = [_channel_last(g.detach().float()) for g in grids]
        Cn = int(texs[0].shape[2])
        if any(int(t.shape[2]) != Cn for t in texs):
            raise RuntimeError("hexplane lookup: all planes of one call must have the same feature width")
        p = pts.detach().contiguous().float()
        lv = levels.detach().contiguous().float()
        N, D = int(p.shape[0]), int(p.shape[1])
        if lv.shape != p.shape:
            raise RuntimeError("hexplane lookup: levels must have the shape of pts")
        ps = _PlaneSet(texs, cols, max_mips, offsets)
        nbytes = L.gsrast_hexplane_scratch_bytes(ps.n, ps.arr, Cn, N)
        if nbytes == 0:
            L.gsrast_hexplane_forward(0, D, Cn, F, ps.n, ps.arr, None, None, None, None, None)     # sets the error text
            raise _lib._err(-1, "gsrast_hexplane_scratch_bytes")
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out = torch.empty((N, F), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = L.gsrast_hexplane_forward(N, D, Cn, F, ps.n, ps.arr, p.data_ptr() if N else None, lv.data_ptr() if N else None,
                                           out.data_ptr() if N else None, scratch.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        if rc != 0:
            raise _lib._err(rc, "gsrast_hexplane_forward")
        ctx.meta = (cols, max_mips, offsets, F, Cn, [tuple(g.shape) for g in grids], [g.dim() for g in grids])
        ctx.save_for_backward(p, lv, scratch, *texs)
        ctx.versions = [g._version for g in grids]
        ctx.grids = None
        return out

    @staticmethod
    def backward(ctx, dout):
        cols, max_mips, offsets, F, Cn, shapes, dims = ctx.meta
        p, lv, scratch, *texs = ctx.saved_tensors
        dev = p.device
        L = _lib.lib()
        N, D = int(p.shape[0]), int(p.shape[1])
        dy = dout.contiguous().float()
        grads = [torch.empty_like(t) for t in texs]
        ps = _PlaneSet(texs, cols, max_mips, offsets, grads)
        need_p, need_l = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        d_pts = torch.empty_like(p) if need_p else None
        d_lv = torch.empty_like(lv) if need_l else None
        with torch.cuda.device(dev):
            rc = L.gsrast_hexplane_backward(N, D, Cn, F, ps.n, ps.arr, p.data_ptr() if N else None, lv.data_ptr() if N else None,
                                            dy.data_ptr() if N else None, d_pts.data_ptr() if need_p and N else None,
                                            d_lv.data_ptr() if need_l and N else None, 1, scratch.data_ptr(),
                                            torch.cuda.current_stream(dev).cuda_stream)
        if rc != 0:
            raise _lib._err(rc, "gsrast_hexplane_backward")
        if N == 0:
            if need_p: d_pts.zero_()
            if need_l: d_lv.zero_()
        # [H,W,C] -> the parameter's [1,C,H,W] as a permuted view (channels_last strides: no copy)
        gout = [g.permute(2, 0, 1).unsqueeze(0) if dims[i] == 4 else g.permute(2, 0, 1) for i, g in enumerate(grads)]
        return (d_pts, d_lv, None, None, None, None, *gout)


def texture_planes(pts: torch.Tensor, levels: torch.Tensor, grids: Sequence[torch.Tensor], cols: Sequence[Sequence[int]],
                   max_mips: Sequence[int], offsets: Sequence[int], feature_width: int) -> torch.Tensor:
    """General entry: plane i samples grids[i] ([1,C,H,W]) at (pts[:, cols[i][0]], pts[:, cols[i][1]]) with the bias
    min(levels[:, cols[i]]) and adds its C channels at features[:, offsets[i]:offsets[i]+C].  Planes sharing an offset must
    be adjacent."""
    return _HexplaneFn.apply(pts, levels, tuple(tuple(c) for c in cols), tuple(max_mips), tuple(offsets), int(feature_width), *grids)


def grid_sample_wrapper(grid: torch.Tensor, coords: torch.Tensor, levels: torch.Tensor, spatio_only: bool, max_level=None,
                        align_corners: bool = True) -> torch.Tensor:
    """scene/hexplane.py:26-60 -- one plane: coords [n,2] (or [1,n,2]), levels [n,2]; returns what the reference's
    `interp.view(B, n, C).squeeze()` gives."""
    if coords.dim() == 3:
        coords = coords[0]
    if coords.shape[-1] != 2:
        raise NotImplementedError(f"Grid-sample was called with {coords.shape[-1]}D data but is only implemented for 2D planes.")
    Cn = grid.shape[-3]
    out = texture_planes(coords, levels.reshape(coords.shape), [grid], [(0, 1)], [7 if spatio_only else 0], [0], Cn)
    return out.view(1, coords.shape[0], Cn).squeeze()


def interpolate_ms_features(pts: torch.Tensor, ms_grids, grid_dimensions: int, concat_features: bool, levels: torch.Tensor,
                            num_levels: Optional[int], concat_planes: bool = False) -> torch.Tensor:
    """scene/hexplane.py:95-139.  pts [N,4] in texture coordinates, ms_grids[scale][plane] = [1,C,H,W] parameters in
    itertools.combinations(range(4), 2) order, levels [N,4].  One fused launch for all scales when the scales are
    concatenated (the configuration ScaleAwareResField fixes, :168); summed scales share one feature block."""
    if grid_dimensions != 2:
        raise NotImplementedError("only 2-D planes (grid_dimensions == 2), as every shipped configuration uses")
    coo = list(itertools.combinations(range(pts.shape[-1]), grid_dimensions))
    if num_levels is None:
        num_levels = len(ms_grids)
    scales = list(ms_grids)[:num_levels]
    pts2 = pts.reshape(-1, pts.shape[-1])
    lv2 = levels.reshape(-1, levels.shape[-1])
    grids: List[torch.Tensor] = []
    cols, mm, offs = [], [], []
    off = 0
    width = 0
    for grid in scales:
        Cn = grid[0].shape[1]
        groups = [[0, 1, 3], [2, 4, 5]] if concat_planes else [list(range(len(coo)))]      # :121-125 space | time blocks
        for gi, members in enumerate(groups):
            for ci in members:
                grids.append(grid[ci]); cols.append(coo[ci]); mm.append(7 if 3 not in coo[ci] else 0)      # :117
                offs.append(off + gi * Cn if concat_features else gi * Cn)
        block = Cn * len(groups)
        if concat_features:
            off += block
        width = max(width, off if concat_features else block)
    if not concat_features:
        # summed scales: planes of all scales share the blocks; keep equal offsets adjacent
        order = sorted(range(len(grids)), key=lambda i: offs[i])
        grids, cols, mm, offs = [grids[i] for i in order], [cols[i] for i in order], [mm[i] for i in order], [offs[i] for i in order]
    return texture_planes(pts2, lv2, grids, cols, mm, offs, width)
