"""Fused photometric loss for the step right after the rasterizer.

Mirror of how the reference builds its image loss (/root/reference/helper_train.py:50-53 on top of
utils/loss_utils.py:18-19 l1_loss and :38-68 ssim, window 11, sigma 1.5, zero padding, mean over
channels and pixels):

    Ll1  = l1_loss(image, gt)
    loss = (1 - lambda_dssim) * Ll1 + lambda_dssim * (1 - ssim(image, gt))

One HIP forward kernel + one HIP backward kernel (libgsrast_hip.so, `gsrast_loss_*` in include/gsrast.h)
replace the five depthwise convolutions and their autograd replay.  No fallback: GPU tensors only.
"""
from __future__ import annotations

import torch

from diff_gaussian_rasterization_ch3 import _C


class _L1DSSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image: torch.Tensor, gt: torch.Tensor, lambda_dssim: float):
        if not image.is_cuda or not gt.is_cuda:
            raise RuntimeError("fused_loss: tensors must be on a GPU (HIP) device; there is no CPU fallback")
        if image.shape != gt.shape or image.dim() != 3:
            raise RuntimeError("fused_loss: image and gt must both be [C, H, W]")
        image = image.contiguous().float()
        gt = gt.contiguous().float()
        Cn, H, W = (int(v) for v in image.shape)
        L = _C.lib()
        scratch = torch.empty(L.gsrast_loss_scratch_bytes(Cn, H, W), dtype=torch.uint8, device=image.device)
        out3 = torch.empty(3, dtype=torch.float32, device=image.device)
        with torch.cuda.device(image.device):
            rc = L.gsrast_loss_forward(Cn, H, W, image.data_ptr(), gt.data_ptr(), float(lambda_dssim), out3.data_ptr(),
                                       scratch.data_ptr(), torch.cuda.current_stream(image.device).cuda_stream)
        if rc != 0:
            raise _C._err(rc, "gsrast_loss_forward")
        ctx.save_for_backward(image, gt, scratch)
        ctx.lambda_dssim = float(lambda_dssim)
        ctx.mark_non_differentiable(out3)
        return out3[0], out3

    @staticmethod
    def backward(ctx, grad_loss, _grad_parts):
        image, gt, scratch = ctx.saved_tensors
        Cn, H, W = (int(v) for v in image.shape)
        grad = torch.empty_like(image)
        g = grad_loss.contiguous().float().reshape(1)
        with torch.cuda.device(image.device):
            rc = _C.lib().gsrast_loss_backward(Cn, H, W, image.data_ptr(), gt.data_ptr(), ctx.lambda_dssim, g.data_ptr(),
                                               scratch.data_ptr(), grad.data_ptr(),
                                               torch.cuda.current_stream(image.device).cuda_stream)
        if rc != 0:
            raise _C._err(rc, "gsrast_loss_backward")
        return grad, None, None


def l1_dssim_loss(image: torch.Tensor, gt: torch.Tensor, lambda_dssim: float = 0.2, return_parts: bool = False):
    """(1 - lambda) * L1 + lambda * (1 - SSIM), differentiable w.r.t. `image`.
    return_parts=True also returns the detached tensor [loss, l1, ssim] (what train.py logs)."""
    loss, parts = _L1DSSIM.apply(image, gt, lambda_dssim)
    return (loss, parts) if return_parts else loss
