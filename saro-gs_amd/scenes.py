"""Synthetic scenes and cameras for tests and bench.py (NumPy only, platform-stable PCG64).

Definitions follow SURVEY.md section 8(d):

* ``synth(P, seed)``  -- Gaussians in the reference's random-init cube ``U(-1.3, 1.3)^3``
  (/root/reference/scene/dataset_readers.py:526), log-uniform anisotropic scales around the
  neighbour spacing (mirrors the kNN init, scene/saro_gaussian.py:187-188), normalised random
  quaternions (saro_gaussian.py:47), sigmoid-normal opacities (saro_gaussian.py:44), SH-3 colours.
* ``camera(k, V, W, H)`` -- look-at cameras on a radius-4 ring, matrices built with the reference's
  conventions: ``world_view_transform = getWorld2View2(R, T)^T`` (scene/cameras.py:90,
  utils/graphics_utils.py:39-50), ``projection = getProjectionMatrix(0.01, 100, fovx, fovy)^T``
  with ``P[2,2] = (zf+zn)/(zf-zn)`` (graphics_utils.py:52-74), ``full_proj = view @ proj`` in
  row-vector form (cameras.py:100), ``camera_center = inverse(view)[3, :3]`` (cameras.py:101).
  tests/test_oracle_golden.py checks these builders against the reference's own functions.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

C0 = 0.28209479177387814


def world_to_view(R: np.ndarray, t: np.ndarray) -> np.ndarray:
    """4x4 world->view matrix (math layout, NOT transposed) for camera rotation R (camera-to-world,
    as stored by the reference's Camera) and translation t.  graphics_utils.py:39-50 with the
    default translate=0, scale=1 (the inverse / re-inverse pair there is then the identity)."""
    Rt = np.zeros((4, 4), dtype=np.float64)
    Rt[:3, :3] = R.T
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    c2w = np.linalg.inv(Rt)
    Rt = np.linalg.inv(c2w)
    return Rt.astype(np.float32)


def projection(znear: float, zfar: float, fovx: float, fovy: float) -> np.ndarray:
    """graphics_utils.py:52-74 (float32 entries like the torch.zeros(4,4) it fills)."""
    ty = math.tan(fovy / 2)
    tx = math.tan(fovx / 2)
    top, right = ty * znear, tx * znear
    bottom, left = -top, -right
    Pm = np.zeros((4, 4), dtype=np.float32)
    Pm[0, 0] = 2.0 * znear / (right - left)
    Pm[1, 1] = 2.0 * znear / (top - bottom)
    Pm[0, 2] = (right + left) / (right - left)
    Pm[1, 2] = (top + bottom) / (top - bottom)
    Pm[3, 2] = 1.0
    Pm[2, 2] = (zfar + znear) / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    return Pm


def camera(k: int, V: int, W: int, H: int, radius: float = 4.0, elev_deg: float = 20.0,
           fovy: float = 0.6911) -> Dict[str, object]:
    """Camera k of V on a ring, looking at the origin (+z forward, +y down like COLMAP)."""
    az = 2.0 * math.pi * k / max(V, 1)
    el = math.radians(elev_deg)
    eye = np.array([radius * math.cos(el) * math.cos(az), -radius * math.sin(el),
                    radius * math.cos(el) * math.sin(az)], dtype=np.float64)
    fwd = -eye / np.linalg.norm(eye)
    up = np.array([0.0, -1.0, 0.0])
    right = np.cross(up, fwd)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], axis=1)  # camera-to-world rotation (columns = camera axes)
    t = -R.T @ eye                             # world->camera translation
    tanfovy = math.tan(0.5 * fovy)
    tanfovx = tanfovy * W / H
    fovx = 2.0 * math.atan(tanfovx)
    view = world_to_view(R, t)                 # math layout
    proj = projection(0.01, 100.0, fovx, fovy)
    viewmatrix = np.ascontiguousarray(view.T)  # stored transposed, cameras.py:90
    projmatrix = np.ascontiguousarray((viewmatrix.astype(np.float32) @ proj.T.astype(np.float32)))
    campos = np.linalg.inv(viewmatrix.astype(np.float64))[3, :3].astype(np.float32)
    return dict(image_height=int(H), image_width=int(W), tanfovx=float(tanfovx), tanfovy=float(tanfovy),
                viewmatrix=viewmatrix.astype(np.float32), projmatrix=projmatrix.astype(np.float32),
                campos=campos, scale_modifier=1.0, prefiltered=False)


def synth(P: int, seed: int = 0, sh_degree: int = 3, scale_mul: float = 1.0) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    means = rng.uniform(-1.3, 1.3, size=(P, 3)).astype(np.float32)
    s = 0.6 * 2.6 * max(P, 1) ** (-1.0 / 3.0) * scale_mul
    scales = np.exp(rng.uniform(math.log(s / 3.0), math.log(3.0 * s), size=(P, 3))).astype(np.float32)
    q = rng.normal(size=(P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = (1.0 / (1.0 + np.exp(-rng.normal(0.0, 2.0, size=(P, 1))))).astype(np.float32)
    M = 16
    shs = np.zeros((P, M, 3), dtype=np.float32)
    shs[:, 0, :] = rng.uniform(-1.77, 1.77, size=(P, 3))
    shs[:, 1:, :] = rng.normal(0.0, 0.1, size=(P, M - 1, 3))
    return dict(means3D=means, scales=scales, rotations=q.astype(np.float32), opacities=opac,
                shs=shs.astype(np.float32), sh_degree=int(sh_degree),
                bg=np.zeros(3, dtype=np.float32))


def synth_shell(P: int, seed: int = 0, sh_degree: int = 3, radius: float = 1.25, thickness: float = 0.02) -> Dict[str, np.ndarray]:
    """A second occlusion regime for the same P and resolutions: Gaussians on a thin spherical SHELL (a surface, like a trained
    scene) instead of a solid cube.  A camera ray crosses two thin layers, so a tile consumes most of its depth-sorted list before
    the transmittance saturates (R_eff ~ R), where the cube's lists are cut after a few per cent (R_eff / R = 7.5 % at 1 M).
    Scales follow the neighbour spacing on the sphere (sqrt(4 pi r^2 / P)), everything else as synth()."""
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(P, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    means = (d * (radius + thickness * rng.normal(size=(P, 1)))).astype(np.float32)
    s = 0.6 * math.sqrt(4.0 * math.pi * radius * radius / max(P, 1))
    scales = np.exp(rng.uniform(math.log(s / 3.0), math.log(3.0 * s), size=(P, 3))).astype(np.float32)
    q = rng.normal(size=(P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = (1.0 / (1.0 + np.exp(-rng.normal(0.0, 2.0, size=(P, 1))))).astype(np.float32)
    M = 16
    shs = np.zeros((P, M, 3), dtype=np.float32)
    shs[:, 0, :] = rng.uniform(-1.77, 1.77, size=(P, 3))
    shs[:, 1:, :] = rng.normal(0.0, 0.1, size=(P, M - 1, 3))
    return dict(means3D=means, scales=scales, rotations=q.astype(np.float32), opacities=opac,
                shs=shs.astype(np.float32), sh_degree=int(sh_degree), bg=np.zeros(3, dtype=np.float32))


def upstream_grad(H: int, W: int, seed: int) -> np.ndarray:
    """dL/dcolor used by bench and parity tests: N(0,1)/(3HW), SURVEY.md 8(d)."""
    rng = np.random.default_rng(seed)
    return (rng.normal(size=(3, H, W)) / (3.0 * H * W)).astype(np.float32)
