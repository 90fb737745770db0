"""An INDEPENDENT differentiable splatting renderer, written from the mathematics of the method (EWA projection of 3-D
Gaussians, front-to-back alpha compositing) in torch float64 with autograd -- test infrastructure, used by
tests/test_oracle_independent.py to check the oracle's forward AND its hand-derived backward formulas against something that
shares no code and no derivation with oracle/gsrast_oracle.c or the HIP kernels: there is no tile list, no per-tile loop, no
hand-written gradient here; gradients come from autograd of the forward below.

What is taken from the reference (cited) is only what DEFINES the function being differentiated:
  * conventions: row-vector matrices stored transposed (scene/cameras.py:90-100), quaternion (r, x, y, z) used as given
    (forward.cu:127-131), p_w = 1 / (w + 1e-7) (forward.cu:195-197), pixel = ((ndc + 1) * S - 1) / 2 (auxiliary.h:41-44);
  * constants: near-plane cull z <= 0.2 (auxiliary.h:154), frustum clamp 1.3 * tanfov (forward.cu:82-87), + 0.3 on the 2-D
    covariance diagonal (forward.cu:110-111), radius = ceil(3 sqrt(lambda_max)), lambda from max(0.1, mid^2 - det)
    (forward.cu:229-232), 16 x 16 tiles with (int) truncation (auxiliary.h:46-56), alpha = min(0.99, o G), skip alpha < 1/255,
    stop before T (1 - alpha) < 1e-4 (forward.cu:343-357), colour = max(SH + 0.5, 0) (forward.cu:60-70);
  * two deliberate deviations of the reference's backward from the true derivative, reproduced so that the comparison is
    meaningful: the 0.99 clamp of alpha passes gradients through (backward.cu:538, :554 apply no mask) -> straight-through
    here (`clamp_passthrough`); the depth output carries no gradient.

Discrete decisions (culling, rectangles, depth order, the alpha / transmittance thresholds) are taken on detached float64
values; `ambiguous` flags every pixel in which one of those decisions sits within fp32 rounding of its threshold -- the
caller zeroes the upstream gradient there and skips those pixels, so an fp32 implementation is never asked to reproduce a coin
flip."""
from __future__ import annotations

import math

import numpy as np
import torch

PI = math.pi
# the function's constants are the reference's fp32 literals (0.3f, 0.99f, 1.0f / 255.0f, 0.0001f, 0.0000001f, 0.2f, 1.3f)
F32 = lambda v: float(np.float32(v))      # noqa: E731
C_DILATE, C_AMAX, C_AMIN, C_TMIN, C_WEPS, C_NEAR, C_LIM = F32(0.3), F32(0.99), float(np.float32(1.0) / np.float32(255.0)), F32(0.0001), F32(0.0000001), F32(0.2), F32(1.3)
# real spherical-harmonics normalisation constants from their closed forms, rounded to fp32 like the kernel's literals
# (auxiliary.h:22-39 holds them as float)
SH_C0 = F32(0.5 * math.sqrt(1.0 / PI))
SH_C1 = F32(math.sqrt(3.0 / (4.0 * PI)))
SH_C2 = tuple(F32(v) for v in (0.5 * math.sqrt(15.0 / PI), -0.5 * math.sqrt(15.0 / PI), 0.25 * math.sqrt(5.0 / PI), -0.5 * math.sqrt(15.0 / PI),
                               0.25 * math.sqrt(15.0 / PI)))
SH_C3 = tuple(F32(v) for v in (-0.25 * math.sqrt(35.0 / (2.0 * PI)), 0.5 * math.sqrt(105.0 / PI), -0.25 * math.sqrt(21.0 / (2.0 * PI)),
                               0.25 * math.sqrt(7.0 / PI), -0.25 * math.sqrt(21.0 / (2.0 * PI)), 0.25 * math.sqrt(105.0 / PI),
                               -0.25 * math.sqrt(35.0 / (2.0 * PI))))


def sh_colour(deg: int, sh: torch.Tensor, d: torch.Tensor) -> torch.Tensor:
    """Real spherical harmonics up to degree 3, sh [P, M, 3], unit directions d [P, 3] -> [P, 3]."""
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    c = SH_C0 * sh[:, 0]
    if deg > 0:
        c = c - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        c = (c + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
             + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
        if deg > 2:
            c = (c + SH_C3[0] * y * (3.0 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                 + SH_C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11] + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
                 + SH_C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                 + SH_C3[6] * x * (xx - 3.0 * yy) * sh[:, 15])
    return c


def rotation_matrix(q: torch.Tensor) -> torch.Tensor:
    """Rotation matrix of the quaternion (r, x, y, z) AS GIVEN (no normalisation)."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]
    return torch.stack(rows, dim=1).reshape(-1, 3, 3)


def project(means3D, scales, rotations, cam, scale_modifier=1.0, cov3D=None, ndc_offset=None):
    """Per-Gaussian screen-space quantities (all differentiable) + the discrete ones (numpy)."""
    W, H = int(cam["image_width"]), int(cam["image_height"])
    V = torch.as_tensor(np.asarray(cam["viewmatrix"], np.float64))            # transposed storage: row vector @ V
    Pm = torch.as_tensor(np.asarray(cam["projmatrix"], np.float64))
    tanx, tany = F32(cam["tanfovx"]), F32(cam["tanfovy"])      # the entry points take float arguments
    fx, fy = W / (2.0 * tanx), H / (2.0 * tany)                # rasterizer_impl.cu:222-223
    ones = torch.ones_like(means3D[:, :1])
    ph = torch.cat([means3D, ones], dim=1)
    t = (ph @ V)[:, :3]                                                       # view space
    hom = ph @ Pm
    pw = 1.0 / (hom[:, 3] + C_WEPS)
    ndc = hom[:, :2] * pw[:, None]
    if ndc_offset is not None:
        ndc = ndc + ndc_offset                                               # gradient sink for the screen-space mean
    pix = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], dim=1)
    if cov3D is None:
        R = rotation_matrix(rotations)
        Mx = R * (scale_modifier * scales)[:, None, :]                        # R @ diag(s)
        Sigma = Mx @ Mx.transpose(1, 2)
    else:
        c = cov3D
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], dim=1).reshape(-1, 3, 3)
    limx, limy = C_LIM * tanx, C_LIM * tany
    tz = t[:, 2]
    txc = torch.clamp(t[:, 0] / tz, -limx, limx) * tz
    tyc = torch.clamp(t[:, 1] / tz, -limy, limy) * tz
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -fx * txc / (tz * tz), zero, fy / tz, -fy * tyc / (tz * tz)], dim=1).reshape(-1, 2, 3)
    Wr = V[:3, :3].T                                                          # world -> view rotation (math layout)
    A = J @ Wr
    cov2 = A @ Sigma @ A.transpose(1, 2)
    a = cov2[:, 0, 0] + C_DILATE
    b = cov2[:, 0, 1]
    c2 = cov2[:, 1, 1] + C_DILATE
    det = a * c2 - b * b
    conic = torch.stack([c2 / det, -b / det, a / det], dim=1)
    # discrete part
    with torch.no_grad():
        mid = 0.5 * (a + c2)
        lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
        r_real = 3.0 * torch.sqrt(lam)
        radius = torch.ceil(r_real)
        vis = (tz > C_NEAR) & (det != 0.0)
        gx, gy = (W + 15) // 16, (H + 15) // 16
        trunc = lambda v: torch.trunc(v)      # noqa: E731  (int) cast
        x0 = torch.clamp(trunc((pix[:, 0] - radius) / 16.0), 0, gx); x1 = torch.clamp(trunc((pix[:, 0] + radius + 15.0) / 16.0), 0, gx)
        y0 = torch.clamp(trunc((pix[:, 1] - radius) / 16.0), 0, gy); y1 = torch.clamp(trunc((pix[:, 1] + radius + 15.0) / 16.0), 0, gy)
        tiles = (x1 - x0) * (y1 - y0)
        vis = vis & (tiles > 0)
        # how close the discrete decisions are to flipping under fp32 rounding of their inputs
        frac = torch.abs(r_real - torch.round(r_real))
        edge = lambda v: torch.abs(v - torch.round(v))      # noqa: E731
        rect_margin = torch.minimum(torch.minimum(edge((pix[:, 0] - radius) / 16.0), edge((pix[:, 0] + radius + 15.0) / 16.0)),
                                    torch.minimum(edge((pix[:, 1] - radius) / 16.0), edge((pix[:, 1] + radius + 15.0) / 16.0)))
    disc = dict(vis=vis.numpy(), radius=(radius * vis).numpy().astype(np.int64), rect=torch.stack([x0, y0, x1, y1], 1).numpy().astype(np.int64),
                tiles=(tiles * vis).numpy().astype(np.int64), radius_margin=frac.numpy(), rect_margin=rect_margin.numpy())
    return dict(pix=pix, conic=conic, depth=tz, cov2=(a, b, c2), Sigma=Sigma, disc=disc, W=W, H=H)


def clamp_passthrough(alpha: torch.Tensor, hi: float) -> torch.Tensor:
    return alpha + (torch.clamp(alpha, max=hi) - alpha).detach()


def render(means3D, scales, rotations, opacities, shs, sh_degree, cam, bg, *, colors_precomp=None, cov3D=None,
           scale_modifier=1.0, ndc_offset=None, fp32_eps=4e-6):
    """Returns dict(color [3,H,W], depth [H,W], final_T [H,W], ambiguous [H,W] bool, proj=...)."""
    pr = project(means3D, scales, rotations, cam, scale_modifier, cov3D, ndc_offset)
    W, H = pr["W"], pr["H"]
    vis = torch.as_tensor(pr["disc"]["vis"])
    if colors_precomp is None:
        campos = torch.as_tensor(np.asarray(cam["campos"], np.float64))
        d = means3D - campos
        d = d / torch.linalg.norm(d, dim=1, keepdim=True)
        raw = sh_colour(int(sh_degree), shs, d) + 0.5
        col = torch.clamp(raw, min=0.0)
        col_margin = raw.detach().abs().min(dim=1).values
    else:
        col = colors_precomp
        col_margin = torch.full((means3D.shape[0],), 1.0, dtype=torch.float64)
    idx = torch.nonzero(vis)[:, 0]
    # depth order as the 64-bit key sort gives it inside any tile: by view-space z, ties by index (stable)
    z = pr["depth"].detach()[idx]
    order = torch.as_tensor(np.lexsort((idx.numpy(), z.numpy())))
    idx = idx[order]
    zs = z[order]
    gap = (zs[1:] - zs[:-1]).min().item() if len(zs) > 1 else 1.0
    pix, conic, o = pr["pix"][idx], pr["conic"][idx], opacities.reshape(-1)[idx]
    colk, depk = col[idx], pr["depth"][idx]
    rect = torch.as_tensor(pr["disc"]["rect"])[idx]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    px, py = xs.reshape(-1, 1), ys.reshape(-1, 1)                            # [N, 1]
    tx, ty = torch.div(px, 16, rounding_mode="floor"), torch.div(py, 16, rounding_mode="floor")
    listed = (tx >= rect[None, :, 0]) & (tx < rect[None, :, 2]) & (ty >= rect[None, :, 1]) & (ty < rect[None, :, 3])   # [N, K]
    dx = pix[None, :, 0] - px
    dy = pix[None, :, 1] - py
    power = -0.5 * (conic[None, :, 0] * dx * dx + conic[None, :, 2] * dy * dy) - conic[None, :, 1] * dx * dy
    G = torch.exp(power)
    alpha_raw = o[None, :] * G
    alpha = clamp_passthrough(alpha_raw, C_AMAX)
    with torch.no_grad():
        ok = listed & (power <= 0.0) & (alpha >= C_AMIN)
    one_minus = torch.where(ok, 1.0 - alpha, torch.ones_like(alpha))
    T_incl = torch.cumprod(one_minus, dim=1)
    T_excl = T_incl / one_minus
    with torch.no_grad():
        live = ok & (T_incl >= C_TMIN)                                         # the pair that would push T below 1e-4 ends the pixel
        # a pixel is finished at the FIRST such pair; later pairs are dropped even though T_incl is monotone anyway
        # ambiguity of the decisions under fp32 rounding (relative eps on alpha and T, absolute on power)
        scale_q = (conic[None, :, 0].abs() * dx * dx + conic[None, :, 2].abs() * dy * dy + 2 * conic[None, :, 1].abs() * (dx * dy).abs())
        amb = listed & ((power.abs() <= fp32_eps * (1.0 + scale_q)) |
                        ((power <= 0.0) & ((alpha_raw / C_AMIN - 1.0).abs() <= 8 * fp32_eps * (1.0 + scale_q))))
        amb = amb | (ok & ((T_incl / C_TMIN - 1.0).abs() <= 64 * fp32_eps))
        T_before_first_dead = T_incl
        amb_pix = amb.any(dim=1)
    w = torch.where(live, alpha * T_excl, torch.zeros_like(alpha))          # blending weights
    colour = w @ colk                                                        # [N, 3]
    T_final = torch.where(live, one_minus, torch.ones_like(alpha)).prod(dim=1)
    bgt = torch.as_tensor(np.asarray(bg, np.float64))
    colour = colour + T_final[:, None] * bgt[None, :]
    with torch.no_grad():
        # median depth: the contributing pair at which T crosses 0.5 (forward.cu:368-372); default 15
        T_after = torch.where(live, T_incl, torch.full_like(T_incl, 2.0))
        T_prev = torch.where(live, T_excl, torch.full_like(T_incl, -1.0))
        cross = live & (T_prev > 0.5) & (T_after < 0.5)
        has = cross.any(dim=1)
        first = torch.argmax(cross.to(torch.int8), dim=1)
        depth = torch.where(has, depk.detach()[first], torch.full((H * W,), 15.0, dtype=torch.float64))
        amb_pix = amb_pix | ((live & (((T_prev - 0.5).abs() <= 64 * fp32_eps) | ((T_after - 0.5).abs() <= 64 * fp32_eps))).any(dim=1))
        n_live = live.sum(dim=1)
    return dict(color=colour.T.reshape(3, H, W), depth=depth.reshape(H, W), final_T=T_final.reshape(H, W),
                ambiguous=amb_pix.reshape(H, W).numpy(), proj=pr, order=idx.numpy(), min_depth_gap=gap,
                colour_clamp_margin=col_margin.numpy(), n_live=n_live.reshape(H, W).numpy(),
                clamped_pairs=int((live & (alpha_raw > C_AMAX)).sum().item()))
