"""Fused L1 + D-SSIM loss (the "next" row after the rasterizer).

CPU: the numpy oracle against an independently written torch restatement (conv2d, fp64) and against the
committed golden vector.  GPU (-m gpu): the HIP kernels through the C ABI against both, forward and backward."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

G = os.path.join(os.path.dirname(__file__), "golden")


def _images(C, H, W, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    base = 0.5 + 0.35 * np.sin(xx / 7.0 + seed) * np.cos(yy / 5.0)
    gt = np.clip(base[None] + 0.1 * rng.normal(size=(C, H, W)), 0, 1).astype(np.float32)
    img = np.clip(gt + 0.08 * rng.normal(size=(C, H, W)) + 0.05 * np.sin(yy / 3.0)[None], 0, 1).astype(np.float32)
    img[:, : H // 6] = gt[:, : H // 6]          # a band of exactly equal pixels: |x-y| has a zero sub-gradient there
    return img, gt


def torch_loss(img: torch.Tensor, gt: torch.Tensor, lam: float):
    """Straight torch (fp64) version of utils/loss_utils.py:38-68 + helper_train.py:50-53, written for this test."""
    C = img.shape[0]
    g = torch.tensor([math.exp(-(x - 5) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)], dtype=torch.float32)
    g = g / g.sum()
    w = (g[:, None] @ g[None, :]).to(img.dtype).expand(C, 1, 11, 11).contiguous()
    conv = lambda a: F.conv2d(a[None], w, padding=5, groups=C)[0]  # noqa: E731
    mu1, mu2 = conv(img), conv(gt)
    s1 = conv(img * img) - mu1 * mu1
    s2 = conv(gt * gt) - mu2 * mu2
    s12 = conv(img * gt) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    smap = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))
    l1 = (img - gt).abs().mean()
    return (1 - lam) * l1 + lam * (1 - smap.mean()), l1, smap.mean()


@pytest.mark.parametrize("C,H,W", [(3, 40, 56), (1, 17, 23), (3, 64, 64)])
def test_numpy_oracle_matches_torch_restatement(C, H, W):
    from oracle import loss_oracle
    img, gt = _images(C, H, W, 3)
    want = torch_loss(torch.from_numpy(img).double(), torch.from_numpy(gt).double(), 0.2)
    got = loss_oracle.l1_dssim(img, gt, 0.2)
    # both build the fp32 window like the reference; the fp32 normalising sum may round differently in
    # numpy and torch (1 ulp of the window = ~1e-7 relative) -- everything else is fp64
    for a, b in zip(got, want):
        assert abs(a - float(b)) < 5e-7


def test_oracle_golden_vector():
    from oracle import loss_oracle
    z = np.load(os.path.join(G, "loss_vectors.npz"))
    got = loss_oracle.l1_dssim(z["img"], z["gt"], float(z["lambda_dssim"]))
    np.testing.assert_allclose(got, z["loss_l1_ssim"], rtol=0, atol=1e-13)
    # identical images: ssim = 1, l1 = 0, loss = 0
    l, l1, ss = loss_oracle.l1_dssim(z["gt"], z["gt"], 0.2)
    assert abs(ss - 1.0) < 1e-12 and l1 == 0.0 and abs(l) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("C,H,W,lam", [(3, 40, 56, 0.2), (3, 97, 83, 0.2), (1, 17, 23, 0.5), (3, 16, 16, 1.0), (3, 270, 480, 0.2)])
def test_fused_loss_forward_backward(C, H, W, lam, gpu):
    import fused_loss
    from oracle import loss_oracle
    img, gt = _images(C, H, W, 5)
    x = torch.from_numpy(img).to(gpu).requires_grad_(True)
    y = torch.from_numpy(gt).to(gpu)
    loss, parts = fused_loss.l1_dssim_loss(x, y, lam, return_parts=True)
    (loss * 3.0).backward()                     # upstream gradient != 1
    want = loss_oracle.l1_dssim(img, gt, lam)
    np.testing.assert_allclose(parts.cpu().numpy(), np.array(want), rtol=0, atol=2e-6)
    assert abs(float(loss.detach()) - want[0]) < 2e-6
    xr = torch.from_numpy(img).double().requires_grad_(True)
    lr, _, _ = torch_loss(xr, torch.from_numpy(gt).double(), lam)
    (lr * 3.0).backward()
    ref = xr.grad.numpy()
    got = x.grad.cpu().numpy().astype(np.float64)
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 2e-5 * scale + 1e-12, (np.abs(got - ref).max(), scale)
    assert not got[:, : H // 6][np.equal(img, gt)[:, : H // 6]].any() or lam > 0     # L1 part is zero where x == y


@pytest.mark.gpu
def test_fused_loss_full_size_properties(gpu):
    """1080p: deterministic, loss(x, x) = 0 with ssim = 1, gradient of a scaled loss scales."""
    import fused_loss
    C, H, W = 3, 1080, 1920
    torch.manual_seed(0)
    y = torch.rand(C, H, W, device=gpu)
    x = (y + 0.05 * torch.randn(C, H, W, device=gpu)).clamp(0, 1).requires_grad_(True)
    l1, p1 = fused_loss.l1_dssim_loss(x, y, 0.2, return_parts=True)
    l2, p2 = fused_loss.l1_dssim_loss(x, y, 0.2, return_parts=True)
    assert torch.equal(p1, p2)
    l1.backward()
    g1 = x.grad.clone(); x.grad = None
    (2.5 * l2).backward()
    assert torch.allclose(x.grad, 2.5 * g1, rtol=1e-6, atol=0)
    same, ps = fused_loss.l1_dssim_loss(y, y, 0.2, return_parts=True)
    assert abs(float(ps[2]) - 1.0) < 1e-6 and float(ps[1]) == 0.0 and abs(float(same)) < 1e-6
    assert 0.0 < float(p1[2]) < 1.0 and float(p1[1]) > 0.0


def test_fused_loss_has_no_cpu_fallback():
    import fused_loss
    with pytest.raises(RuntimeError, match="GPU"):
        fused_loss.l1_dssim_loss(torch.zeros(3, 8, 8), torch.zeros(3, 8, 8))
