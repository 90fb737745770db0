"""CPU restatement of the reference's photometric loss (TEST INFRASTRUCTURE ONLY -- never imported by
the product package).

Follows /root/reference/utils/loss_utils.py:
  :18-19  l1_loss          mean |x - y|
  :25-35  window           11 taps, exp(-(i-5)^2 / (2*1.5^2)) evaluated in double, stored fp32, divided by its fp32
                           sum; the 2-D window is the fp32 outer product
  :48-68  _ssim            five depthwise convolutions with zero padding 5; C1 = 0.01^2, C2 = 0.03^2; mean
and /root/reference/helper_train.py:50-53:  loss = (1 - lambda) * Ll1 + lambda * (1 - ssim).

Pinning: PINNED against the reference itself -- tests/golden/make_golden.py imports utils/loss_utils.py (with an inert
placeholder for its unused module-level `torchmetrics` import) and stores the window, l1_loss, ssim, the combined loss and its
autograd gradient for three image pairs (tests/golden/loss_vectors.npz, keys ref_*); tests/test_oracle_golden.py checks this
file against them.  tests/test_loss.py additionally cross-checks it against an independently written torch (conv2d +
autograd, fp64) version of the same formulas.
All arithmetic here is float64 on the fp32 window values.
"""
from __future__ import annotations

import math

import numpy as np


def window2d() -> np.ndarray:
    g = np.array([math.exp(-(x - 5) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)], dtype=np.float32)
    # gauss / gauss.sum() (loss_utils.py:27): torch's CPU sum of these 11 floats is the correctly rounded one (a sequential or
    # numpy pairwise fp32 sum lands 1 ulp lower) -- pinned by the reference's own window in tests/golden/loss_vectors.npz
    g = (g / np.float32(g.astype(np.float64).sum())).astype(np.float32)
    return np.outer(g, g).astype(np.float32).astype(np.float64)


def _blur(a: np.ndarray, w: np.ndarray) -> np.ndarray:
    """Depthwise 11x11 correlation with zero padding 5; a is [C, H, W] float64."""
    C, H, W = a.shape
    p = np.zeros((C, H + 10, W + 10), dtype=np.float64)
    p[:, 5:5 + H, 5:5 + W] = a
    out = np.zeros_like(a)
    for i in range(11):
        for j in range(11):
            out += w[i, j] * p[:, i:i + H, j:j + W]
    return out


def ssim_map(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    x = x.astype(np.float64); y = y.astype(np.float64)
    w = window2d()
    mu1, mu2 = _blur(x, w), _blur(y, w)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = _blur(x * x, w) - mu1_sq
    s2 = _blur(y * y, w) - mu2_sq
    s12 = _blur(x * y, w) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))


def l1_dssim(x: np.ndarray, y: np.ndarray, lambda_dssim: float = 0.2):
    """Returns (loss, l1, ssim) as Python floats."""
    l1 = float(np.abs(x.astype(np.float64) - y.astype(np.float64)).mean())
    ss = float(ssim_map(x, y).mean())
    return (1.0 - lambda_dssim) * l1 + lambda_dssim * (1.0 - ss), l1, ss
