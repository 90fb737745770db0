"""Adam for the per-Gaussian parameter groups with a per-Gaussian learning rate, one HIP launch per step
(third item).

Mirror of how the reference drives its optimizer (/root/reference/scene/saro_gaussian.py):
  :306-323  groups xyz / f_dc / f_rest / opacity / scaling / rotation / temporal_pos, torch.optim.Adam(l, lr=0.0, eps=1e-15)
  :345-398  update_learning_rate assigns  param_group['lr'] = lr * self.inv_intergral  -- a [P,1] tensor: one rate per row

`GaussianAdam` keeps torch.optim.Adam's surface for those groups (`param_groups` with 'params' / 'lr' / 'name',
`step()`, `zero_grad()`, `state`), accepts a float or a [P] / [P,1] tensor as a group's 'lr', and updates every group in ONE
kernel (`gsrast_adam_step`).  amsgrad / maximize / weight_decay are not supported (the reference does not use them for
these groups).  GPU tensors only; no fallback.
"""
from __future__ import annotations

from typing import Dict, Iterable, List

import torch

from diff_gaussian_rasterization_ch3 import _C


class GaussianAdam:
    def __init__(self, param_groups: Iterable[Dict], betas=(0.9, 0.999), eps: float = 1e-15):
        self.param_groups: List[Dict] = []
        for g in param_groups:
            g = dict(g)
            ps = g["params"]
            g["params"] = [ps] if isinstance(ps, torch.Tensor) else list(ps)
            if len(g["params"]) != 1:
                raise ValueError("GaussianAdam: one tensor per group (the per-Gaussian groups of saro_gaussian.py:306-318)")
            g.setdefault("lr", 0.0)
            self.param_groups.append(g)
        if len(self.param_groups) > 8:
            raise ValueError("GaussianAdam: at most 8 groups per launch")
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)
        self.state: Dict[torch.Tensor, Dict[str, torch.Tensor]] = {}
        self._step = 0

    def zero_grad(self, set_to_none: bool = True) -> None:
        for g in self.param_groups:
            p = g["params"][0]
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self) -> None:
        L = _C.lib()
        arr = (_C.AdamGroupStruct * len(self.param_groups))()
        n, keep, dev = 0, [], None
        for g in self.param_groups:
            p = g["params"][0]
            if p.grad is None:
                continue
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("GaussianAdam: parameters must be contiguous float32 GPU tensors (no CPU fallback)")
            dev = p.device
            st = self.state.get(p)
            if st is None:
                st = self.state[p] = {"exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
            grad = p.grad.contiguous()
            rows = int(p.shape[0]) if p.dim() > 0 else 1
            width = p.numel() // max(rows, 1) if rows else 1
            lr = g["lr"]
            lr_rows = None
            if isinstance(lr, torch.Tensor) and lr.numel() > 1:
                if lr.numel() != rows:
                    raise RuntimeError(f"GaussianAdam: per-row lr of group {g.get('name')} has {lr.numel()} entries, the parameter {rows} rows")
                lr_rows = lr.detach().to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
                lr_scalar = 1.0
            else:
                lr_scalar = float(lr)
            a = arr[n]
            a.param, a.grad, a.exp_avg, a.exp_avg_sq = p.data_ptr(), grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            a.lr_rows = lr_rows.data_ptr() if lr_rows is not None else None
            a.lr, a.rows, a.width = lr_scalar, rows, max(width, 1)
            keep.extend((grad, lr_rows))
            n += 1
        self._step += 1
        if n == 0:
            return
        with torch.cuda.device(dev):
            rc = L.gsrast_adam_step(n, arr, self.betas[0], self.betas[1], self.eps, self._step, torch.cuda.current_stream(dev).cuda_stream)
        if rc != 0:
            raise _C._err(rc, "gsrast_adam_step")
