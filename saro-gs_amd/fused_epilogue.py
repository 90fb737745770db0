"""Fused activation / deformation epilogue: the step right BEFORE the rasterizer.

Mirror of the tail of get_deformation (/root/reference/scene/saro_gaussian.py:807-847) with the activations of
saro_gaussian.py:39-47:

    motion  = _xyz + motion_residual                                   # args.dx
    rot     = normalize(_rotation + rot_residual[:, :4])               # args.drot
    scale   = exp(_scaling + rot_residual[:, 4:])
    opacity = sigmoid(_opacity) * trbfoutput                           # args.dopacity
    shs     = cat(_features_dc, _features_rest, dim=1) + shs_residual  # args.dsh

Every residual (and trbfoutput) is optional: with all of them None this is the static stage's plain activations
(get_scaling / get_rotation / get_opacity / get_features).  Two HIP kernels forward, one backward
(libgsrast_hip.so, `gsrast_activate_*` in include/gsrast.h); the SH part of the backward is views of dL/dshs.
No fallback: GPU tensors only.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from diff_gaussian_rasterization_ch3 import _C


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class _Activate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, motion_res, rotation, rot_res, scaling, opacity, trbf, f_dc, f_rest, shs_res):
        if not xyz.is_cuda:
            raise RuntimeError("fused_epilogue: tensors must be on a GPU (HIP) device; there is no CPU fallback")
        dev = xyz.device
        c = lambda t: None if t is None else t.contiguous().float()  # noqa: E731
        xyz, motion_res, rotation, rot_res, scaling = c(xyz), c(motion_res), c(rotation), c(rot_res), c(scaling)
        opacity, trbf, f_dc, f_rest, shs_res = c(opacity), c(trbf), c(f_dc), c(f_rest), c(shs_res)
        P = int(xyz.shape[0])
        M = 1 + int(f_rest.shape[1])
        if rot_res is not None and tuple(rot_res.shape) != (P, 7):
            raise RuntimeError("fused_epilogue: rot_residual must be [P, 7] (4 rotation + 3 scale)")
        if f_dc.shape != (P, 1, 3) or f_rest.shape[0] != P or f_rest.shape[2] != 3:
            raise RuntimeError("fused_epilogue: features_dc must be [P,1,3], features_rest [P,M-1,3]")
        o = dict(dtype=torch.float32, device=dev)
        motion, rot, scale = torch.empty((P, 3), **o), torch.empty((P, 4), **o), torch.empty((P, 3), **o)
        opa, shs = torch.empty((P, 1), **o), torch.empty((P, M, 3), **o)
        with torch.cuda.device(dev):
            rc = _C.lib().gsrast_activate_forward(
                P, M, _p(xyz), _p(motion_res), _p(rotation), _p(rot_res), _p(scaling), _p(opacity), _p(trbf), _p(f_dc),
                _p(f_rest), _p(shs_res), _p(motion), _p(rot), _p(scale), _p(opa), _p(shs),
                torch.cuda.current_stream(dev).cuda_stream)
        if rc != 0:
            raise _C._err(rc, "gsrast_activate_forward")
        ctx.save_for_backward(rotation, rot_res, scale, opacity, trbf)
        ctx.has = (motion_res is not None, rot_res is not None, trbf is not None, shs_res is not None)
        ctx.set_materialize_grads(False)
        return motion, rot, scale, opa, shs

    @staticmethod
    def backward(ctx, d_motion, d_rot, d_scale, d_opa, d_shs):
        rotation, rot_res, scale, opacity, trbf = ctx.saved_tensors
        has_mres, has_rres, has_trbf, has_sres = ctx.has
        dev = rotation.device
        P = int(rotation.shape[0])
        o = dict(dtype=torch.float32, device=dev)
        c = lambda t: None if t is None else t.contiguous().float()  # noqa: E731
        d_rot, d_scale, d_opa = c(d_rot), c(d_scale), c(d_opa)
        g_rotation, g_scaling, g_logit = torch.empty((P, 4), **o), torch.empty((P, 3), **o), torch.empty((P, 1), **o)
        g_rres = torch.empty((P, 7), **o) if has_rres else None
        g_trbf = torch.empty((P, 1), **o) if has_trbf else None
        with torch.cuda.device(dev):
            rc = _C.lib().gsrast_activate_backward(
                P, _p(rotation), _p(rot_res), _p(scale), _p(opacity), _p(trbf), _p(d_rot), _p(d_scale), _p(d_opa),
                _p(g_rotation), _p(g_scaling), _p(g_rres), _p(g_logit), _p(g_trbf), torch.cuda.current_stream(dev).cuda_stream)
        if rc != 0:
            raise _C._err(rc, "gsrast_activate_backward")
        g_dc = g_rest = g_sres = None
        if d_shs is not None:           # no kernel: slices / the tensor itself
            g_dc, g_rest = d_shs[:, :1, :], d_shs[:, 1:, :]
            g_sres = d_shs if has_sres else None
        return (d_motion, d_motion if has_mres else None, g_rotation, g_rres, g_scaling, g_logit, g_trbf, g_dc, g_rest, g_sres)


def activate_gaussians(xyz: torch.Tensor, rotation: torch.Tensor, scaling: torch.Tensor, opacity: torch.Tensor,
                       features_dc: torch.Tensor, features_rest: torch.Tensor, *,
                       motion_residual: Optional[torch.Tensor] = None, rot_residual: Optional[torch.Tensor] = None,
                       trbfoutput: Optional[torch.Tensor] = None, shs_residual: Optional[torch.Tensor] = None
                       ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Returns (motion [P,3], rot [P,4], scale [P,3], opacity [P,1], shs [P,M,3]) -- the `means3D, rotations, scales,
    opacities, shs` arguments of GaussianRasterizer -- differentiable w.r.t. every input."""
    return _Activate.apply(xyz, motion_residual, rotation, rot_residual, scaling, opacity, trbfoutput, features_dc,
                           features_rest, shs_residual)
