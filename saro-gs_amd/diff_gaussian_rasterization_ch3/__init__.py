"""diff_gaussian_rasterization_ch3 -- drop-in replacement, MI355X (gfx950) native.

Same import name and public surface as the rasterizer the SaRO-GS renderer imports
(/root/reference/renderer/__init__.py:32):

    from diff_gaussian_rasterization_ch3 import GaussianRasterizationSettings, GaussianRasterizer

so ``renderer.train_render`` / ``test_render`` and ``scene/saro_gaussian.py`` can call it unchanged.
Behaviour follows /root/reference/submodules/gaussian_rasterization_ch3/
diff_gaussian_rasterization_ch3/__init__.py (cited below as REF:line):

* ``GaussianRasterizationSettings`` -- NamedTuple with the reference's eleven fields in the
  reference's order (REF:134-145).
* ``GaussianRasterizer(raster_settings)(means3D, means2D, opacities, shs=None, colors_precomp=None,
  scales=None, rotations=None, cov3D_precomp=None)`` -> ``(color[3,H,W], radii[P] int32,
  depth[1,H,W])`` (REF:163-196, REF:85).  Exactly one of (shs, colors_precomp) and exactly one of
  ((scales, rotations), cov3D_precomp) must be given, otherwise ``Exception`` (REF:167-171).
* autograd: gradients flow to means3D, means2D (a [P,3] tensor whose [:, :2] drives densification),
  shs / colors_precomp, opacities, scales, rotations, cov3D_precomp; depth and radii carry no
  gradient (REF:88, REF:120-130).
* ``GaussianRasterizer.markVisible(positions)`` (REF:152-161).

The compute is in ``libgsrast_hip.so`` (hand-written HIP kernels behind the C ABI of
``include/gsrast.h``), reached through ``_C`` (ctypes).  There is no CPU / PyTorch fallback.
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _C

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool


class _RasterizeGaussians(torch.autograd.Function):
    """Opaque-state autograd node: forward saves the three state buffers the native library
    filled, backward hands them back (REF:42-132)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        rs = raster_settings
        ar = _C._grad_arena
        if ar is not None and ar.sh_factors and sh.numel() != 0 and sh.requires_grad and not sh.is_leaf:
            # factor mode completes shs.grad later (sh_grad_combine writes the arena): that only reaches the parameters if
            # shs itself is the leaf.  cat(features_dc, features_rest) (get_features) would copy the unfinished buffer.
            raise RuntimeError("GradArena(sh_factors=True) needs the rasterizer's `shs` to be a leaf tensor; for "
                               "cat(features_dc, features_rest) or shs + residual use GradArena(sh_factors=False) + "
                               "view_parallel.allreduce_mean_inplace")
        num_rendered, color, radii, geom_buf, bin_buf, img_buf, depth = _C.rasterize_gaussians(
            rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
            rs.sh_degree, rs.campos, rs.prefiltered, forward_only=not any(ctx.needs_input_grad))
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.gs_options = _C.current_options()      # the backward runs on autograd's thread: it must use THIS thread's options
        ctx.gs_options["forward_only"] = int(not any(ctx.needs_input_grad))   # (a backward then cannot happen; kept consistent anyway)
        ctx.gs_backwards = 0                       # backwards run on this state (retain_graph): only the first finds zeroed records
        # opacities are not saved: the state buffer keeps them next to the conic (REF:84)
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh,
                              geom_buf, bin_buf, img_buf)
        # depth stays "differentiable" as in the reference (REF:85-88: it is returned by the Function, its incoming gradient is ignored): a
        # loss built from depth alone runs a backward that yields zero gradients there, and does here (round 6; rounds 1-5 marked it
        # non-differentiable, which raised instead).  radii is int32: never differentiable.
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)     # no zero-filled [1,H,W] / [P] gradients for the two outputs nothing flows through
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii, _grad_depth):
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh,
         geom_buf, bin_buf, img_buf) = ctx.saved_tensors
        if grad_out_color is None:      # a loss that reaches this node through depth only: the reference sees a zero colour gradient (REF:88)
            grad_out_color = torch.zeros((_C.NUM_CHANNELS, rs.image_height, rs.image_width), device=means3D.device)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
         grad_scales, grad_rotations) = _C.rasterize_gaussians_backward(
            rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, sh, rs.sh_degree, rs.campos,
            geom_buf, ctx.num_rendered, bin_buf, img_buf, options=ctx.gs_options, first_backward=ctx.gs_backwards == 0)
        ctx.gs_backwards += 1
        # one gradient per forward input, in input order; absent optionals get None
        def opt(g, x):
            return g if x.numel() != 0 else None
        return (grad_means3D, grad_means2D, opt(grad_sh, sh), opt(grad_colors_precomp, colors_precomp),
                grad_opacities, opt(grad_scales, scales), opt(grad_rotations, rotations),
                opt(grad_cov3Ds_precomp, cov3Ds_precomp), None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    """Functional form (REF:17-39)."""
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


_EMPTY = torch.empty(0)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """bool[P]: Gaussians in front of the near plane (view-space z > 0.2)."""
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs: Optional[torch.Tensor] = None,
                colors_precomp: Optional[torch.Tensor] = None, scales: Optional[torch.Tensor] = None,
                rotations: Optional[torch.Tensor] = None, cov3D_precomp: Optional[torch.Tensor] = None):
        have_sh, have_rgb = shs is not None, colors_precomp is not None
        if have_sh == have_rgb:
            raise Exception("Please provide exactly one of either SHs or precomputed colors!")
        have_sr = scales is not None or rotations is not None
        full_sr = scales is not None and rotations is not None
        have_cov = cov3D_precomp is not None
        if (not full_sr and not have_cov) or (have_sr and have_cov):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")

        empty = _EMPTY          # absent optional input, like the reference's torch.Tensor([]) (REF:173-183); one shared CPU tensor: nobody writes it
        return rasterize_gaussians(
            means3D, means2D,
            shs if have_sh else empty,
            colors_precomp if have_rgb else empty,
            opacities,
            scales if scales is not None else empty,
            rotations if rotations is not None else empty,
            cov3D_precomp if have_cov else empty,
            self.raster_settings)


# ---- raw-parameter module (no counterpart in the reference: SURVEY.md 8f rank 3 as written -- the activation / deformation epilogue
# of scene/saro_gaussian.py:807-847 fused into the per-Gaussian kernels) -------------------------------------------------------------
class _RasterizeGaussiansRaw(torch.autograd.Function):
    """Inputs in _C.RAW_NAMES order (absent residuals: None) + means2D (the gradient sink of REF:42) + the settings."""

    @staticmethod
    def forward(ctx, means2D, raster_settings, *raw_tensors):
        rs = raster_settings
        raw = dict(zip(_C.RAW_NAMES, raw_tensors))
        forward_only = not any(ctx.needs_input_grad)
        num_rendered, color, radii, geom_buf, bin_buf, img_buf, depth = _C.rasterize_gaussians_raw(
            rs.bg, raw, rs.scale_modifier, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width,
            rs.sh_degree, rs.campos, forward_only=forward_only)
        ctx.raster_settings, ctx.num_rendered = rs, num_rendered
        ctx.gs_options = _C.current_options()
        ctx.gs_options["forward_only"] = int(forward_only)
        ctx.gs_backwards = 0
        ctx.present = tuple(t is not None for t in raw_tensors)
        ctx.save_for_backward(*[t for t in raw_tensors if t is not None], radii, geom_buf, bin_buf, img_buf)
        ctx.mark_non_differentiable(radii)      # (depth: as _RasterizeGaussians -- differentiable in name, its gradient ignored)
        ctx.set_materialize_grads(False)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii, _grad_depth):
        rs = ctx.raster_settings
        saved = list(ctx.saved_tensors)
        img_buf, bin_buf, geom_buf, radii = saved.pop(), saved.pop(), saved.pop(), saved.pop()
        it = iter(saved)
        raw = {n: (next(it) if here else None) for n, here in zip(_C.RAW_NAMES, ctx.present)}
        if grad_out_color is None:
            grad_out_color = torch.zeros((_C.NUM_CHANNELS, rs.image_height, rs.image_width), device=radii.device)
        g = _C.rasterize_gaussians_raw_backward(
            rs.bg, raw, radii, rs.scale_modifier, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, rs.sh_degree,
            rs.campos, geom_buf, ctx.num_rendered, bin_buf, img_buf, options=ctx.gs_options, first_backward=ctx.gs_backwards == 0)
        ctx.gs_backwards += 1
        shapes = {n: (None if raw[n] is None else raw[n].shape) for n in _C.RAW_NAMES}
        grads = tuple(None if raw[n] is None else g[n].reshape(shapes[n]) if g[n].is_contiguous() else g[n] for n in _C.RAW_NAMES)
        return (g["dL_dmeans2D"], None) + grads


class GaussianRasterizerRaw(nn.Module):
    """GaussianRasterizer for callers that hold SaRO-GS's RAW parameters: forward(xyz, means2D, rotation, scaling, opacity, features_dc,
    features_rest, motion_residual=None, rot_residual=None, trbfoutput=None, shs_residual=None) renders
    means3D = xyz + motion_residual, rotations = normalize(rotation + rot_residual[:, :4]), scales = exp(scaling + rot_residual[:, 4:]),
    opacities = sigmoid(opacity) * trbfoutput, shs = cat(features_dc, features_rest) + shs_residual (scene/saro_gaussian.py:807-847) --
    outputs bit-identical to fused_epilogue.activate_gaussians followed by GaussianRasterizer, without the activated tensors ever
    being written.  Gradients flow to every tensor given."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, xyz, means2D, rotation, scaling, opacity, features_dc, features_rest, motion_residual=None, rot_residual=None,
                trbfoutput=None, shs_residual=None):
        raw = dict(xyz=xyz, motion_res=motion_residual, rotation=rotation, rot_res=rot_residual, scaling=scaling, opacity_logit=opacity,
                   trbf=trbfoutput, features_dc=features_dc, features_rest=features_rest, shs_res=shs_residual)
        return _RasterizeGaussiansRaw.apply(means2D, self.raster_settings, *[raw[n] for n in _C.RAW_NAMES])
