"""fp32 Linear layers for the deformation heads with the weight / bias gradient on the HIP library (beyond SURVEY.md 8f; the
measured bottleneck of the dynamic-stage iteration).

The reference builds its heads as `nn.Sequential(nn.Linear, nn.ReLU, nn.Linear, nn.ReLU, nn.Linear[, nn.Sigmoid])`
(/root/reference/scene/saro_gaussian.py:104-110) and evaluates them for every Gaussian (:779-812).  `SplitKLinear` is an
`nn.Linear` (same parameters, same state-dict keys, same forward through the BLAS library) whose backward computes
dW = grad_outᵀ·input and db = Σ grad_out with `gsrast_linear_wgrad` (csrc/gsrast_mlp.h): at 1e6 rows the library runs that
product -- K = 1e6, a 128x128 result -- on 16 workgroups.  `convert_heads(module)` swaps the Linear layers of an existing
head in place.  fp32 throughout, like the reference.  No CPU fallback for the custom backward: CPU tensors take nn.Linear's path."""
import torch
import torch.nn as nn

from diff_gaussian_rasterization_ch3 import _C as _lib


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.addmm(bias, x, weight.t()) if bias is not None else x @ weight.t()

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        dx = g @ weight if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            M, N1, N2 = int(x.shape[0]), int(weight.shape[0]), int(weight.shape[1])
            xc = x.contiguous()
            dw = torch.empty_like(weight, memory_format=torch.contiguous_format)
            db = torch.empty((N1,), dtype=torch.float32, device=x.device) if ctx.has_bias else None
            with torch.cuda.device(x.device):
                rc = _lib.lib().gsrast_linear_wgrad(M, N1, N2, g.data_ptr() if M else None, xc.data_ptr() if M else None, dw.data_ptr(),
                                                    db.data_ptr() if db is not None else None, 0,
                                                    torch.cuda.current_stream(x.device).cuda_stream)
            if rc != 0:
                raise _lib._err(rc, "gsrast_linear_wgrad")
        return dx, dw, db


class SplitKLinear(nn.Linear):
    """nn.Linear with the split-K matrix-core weight gradient for 2-D fp32 GPU inputs of width <= 128."""

    def forward(self, x):
        if (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and self.weight.dtype == torch.float32
                and self.in_features <= 128 and self.out_features <= 128):
            return _LinearFn.apply(x, self.weight, self.bias)
        return super().forward(x)


def convert_heads(module: nn.Module) -> nn.Module:
    """Replace every nn.Linear inside `module` (e.g. motion_mlp / rot_mlp / shs_mlp / opacity_mlp) by a SplitKLinear sharing its
    parameters.  State-dict keys are unchanged."""
    for name, child in list(module.named_children()):
        if type(child) is nn.Linear:
            new = SplitKLinear(child.in_features, child.out_features, bias=child.bias is not None, device=child.weight.device, dtype=child.weight.dtype)
            new.weight, new.bias = child.weight, child.bias
            setattr(module, name, new)
        else:
            convert_heads(child)
    return module
