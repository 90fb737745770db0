"""TEST INFRASTRUCTURE ONLY (see oracle/gsrast_oracle.c for the rules): numpy fp64 restatement of the activation /
deformation epilogue, /root/reference/scene/saro_gaussian.py:807-847 with the activations of :39-47
(exp, sigmoid, F.normalize with eps 1e-12), and of its backward (hand-derived; tests cross-check it against torch
autograd of the reference's own formulation).  Parity unpinned against the reference binary: scene/saro_gaussian.py
imports simple_knn / the CUDA rasterizer at module level, so it cannot be imported here."""
import numpy as np


def forward(xyz, rotation, scaling, opacity, f_dc, f_rest, motion_res=None, rot_res=None, trbf=None, shs_res=None):
    f = lambda a: None if a is None else np.asarray(a, np.float64)  # noqa: E731
    xyz, rotation, scaling, opacity, f_dc, f_rest = map(f, (xyz, rotation, scaling, opacity, f_dc, f_rest))
    motion_res, rot_res, trbf, shs_res = map(f, (motion_res, rot_res, trbf, shs_res))
    motion = xyz + (motion_res if motion_res is not None else 0.0)
    x = rotation + (rot_res[:, :4] if rot_res is not None else 0.0)
    n = np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12)
    rot = x / n
    scale = np.exp(scaling + (rot_res[:, 4:] if rot_res is not None else 0.0))
    s = 1.0 / (1.0 + np.exp(-opacity))
    opa = s * (trbf if trbf is not None else 1.0)
    shs = np.concatenate([f_dc, f_rest], axis=1) + (shs_res if shs_res is not None else 0.0)
    return dict(motion=motion, rot=rot, scale=scale, opacity=opa, shs=shs)


def backward(rotation, scaling, opacity, rot_res, trbf, d_rot, d_scale, d_opa):
    f = lambda a: None if a is None else np.asarray(a, np.float64)  # noqa: E731
    rotation, scaling, opacity, rot_res, trbf, d_rot, d_scale, d_opa = map(f, (rotation, scaling, opacity, rot_res, trbf, d_rot, d_scale, d_opa))
    x = rotation + (rot_res[:, :4] if rot_res is not None else 0.0)
    n = np.linalg.norm(x, axis=1, keepdims=True)
    y = x / np.maximum(n, 1e-12)
    g_rot = np.where(n >= 1e-12, (d_rot - y * (y * d_rot).sum(1, keepdims=True)) / np.maximum(n, 1e-12), d_rot / 1e-12)
    scale = np.exp(scaling + (rot_res[:, 4:] if rot_res is not None else 0.0))
    g_scaling = d_scale * scale
    s = 1.0 / (1.0 + np.exp(-opacity))
    tb = trbf if trbf is not None else 1.0
    return dict(rotation=g_rot, scaling=g_scaling, logit=d_opa * tb * s * (1 - s), trbf=d_opa * s)
