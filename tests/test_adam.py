"""Fused per-row-LR Adam (third item).  CPU: the numpy oracle against torch.optim.Adam.
GPU (-m gpu): the HIP kernel against the oracle (per-row lr) and against torch.optim.Adam on the same device (scalar lr)."""
import numpy as np
import pytest
import torch

SHAPES = {"xyz": (3,), "f_dc": (1, 3), "f_rest": (15, 3), "opacity": (1,), "scaling": (3,), "rotation": (4,), "temporal_pos": (1,)}
LRS = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 1.25e-4, "opacity": 5e-2, "scaling": 5e-3, "rotation": 1e-3, "temporal_pos": 1e-4}


def _data(P, seed, steps):
    rng = np.random.default_rng(seed)
    params = {k: rng.normal(size=(P,) + s).astype(np.float32) for k, s in SHAPES.items()}
    grads = [{k: (rng.normal(size=(P,) + s) * 10.0 ** rng.uniform(-6, 0)).astype(np.float32) for k, s in SHAPES.items()} for _ in range(steps)]
    return params, grads


def test_numpy_oracle_matches_torch_adam():
    from oracle import adam_oracle
    P, steps = 300, 4
    params, grads = _data(P, 1, steps)
    tp = {k: torch.from_numpy(v.copy()).double().requires_grad_(True) for k, v in params.items()}
    opt = torch.optim.Adam([{"params": [tp[k]], "lr": LRS[k], "name": k} for k in SHAPES], lr=0.0, eps=1e-15)
    st = {k: (params[k].astype(np.float64), np.zeros_like(params[k], np.float64), np.zeros_like(params[k], np.float64)) for k in SHAPES}
    for t in range(steps):
        for k in SHAPES:
            tp[k].grad = torch.from_numpy(grads[t][k]).double()
            st[k] = adam_oracle.step(st[k][0], grads[t][k], st[k][1], st[k][2], LRS[k], t + 1)
        opt.step()
    for k in SHAPES:
        np.testing.assert_allclose(st[k][0], tp[k].detach().numpy(), rtol=1e-12, atol=1e-14, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("P", [1000, 4099])
def test_hip_adam_per_row_lr(P, gpu):
    from oracle import adam_oracle
    import fused_adam
    steps = 5
    params, grads = _data(P, 2, steps)
    rng = np.random.default_rng(3)
    inv = (1.0 + 4.0 * rng.random(P)).astype(np.float32)              # inv_intergral / its minimum: >= 1
    gp = {k: torch.from_numpy(v.copy()).to(gpu).requires_grad_(True) for k, v in params.items()}
    per_row = {"xyz", "opacity", "scaling", "rotation", "f_dc", "temporal_pos"}            # saro_gaussian.py:366-398
    opt = fused_adam.GaussianAdam([{"params": [gp[k]], "lr": 0.0, "name": k} for k in SHAPES], eps=1e-15)
    for grp in opt.param_groups:
        k = grp["name"]
        grp["lr"] = LRS[k] * torch.from_numpy(inv).to(gpu).reshape(P, 1) if k in per_row else LRS[k]
    st = {k: (params[k].astype(np.float64), np.zeros_like(params[k], np.float64), np.zeros_like(params[k], np.float64)) for k in SHAPES}
    for t in range(steps):
        for k in SHAPES:
            gp[k].grad = torch.from_numpy(grads[t][k]).to(gpu)
            lr = LRS[k] * inv.astype(np.float64) if k in per_row else LRS[k]
            st[k] = adam_oracle.step(st[k][0], grads[t][k], st[k][1], st[k][2], lr, t + 1)
        opt.step()
    for k in SHAPES:
        got = gp[k].detach().cpu().numpy().astype(np.float64)
        # fp32 parameters vs the fp64 oracle: each of the 5 updates rounds p to fp32 (ulp(|p| <= 4) = 2.4e-7 ... 4.8e-7)
        ref_disp = st[k][0] - params[k]
        np.testing.assert_allclose(got - params[k], ref_disp, rtol=2e-4, atol=1.5e-6, err_msg=k)
        # moments: fp32 sums of terms of mixed sign / magnitude -- absolute error scales with the largest term
        np.testing.assert_allclose(opt.state[gp[k]]["exp_avg"].cpu().numpy(), st[k][1], rtol=1e-5, atol=1e-6 * np.abs(st[k][1]).max(), err_msg=k)
        np.testing.assert_allclose(opt.state[gp[k]]["exp_avg_sq"].cpu().numpy(), st[k][2], rtol=1e-5, atol=1e-6 * np.abs(st[k][2]).max(), err_msg=k)


@pytest.mark.gpu
def test_hip_adam_matches_torch_adam_on_device(gpu):
    import fused_adam
    P, steps = 2000, 6
    params, grads = _data(P, 4, steps)
    a = {k: torch.from_numpy(v.copy()).to(gpu).requires_grad_(True) for k, v in params.items()}
    b = {k: torch.from_numpy(v.copy()).to(gpu).requires_grad_(True) for k, v in params.items()}
    mine = fused_adam.GaussianAdam([{"params": [a[k]], "lr": LRS[k], "name": k} for k in SHAPES], eps=1e-15)
    ref = torch.optim.Adam([{"params": [b[k]], "lr": LRS[k], "name": k} for k in SHAPES], lr=0.0, eps=1e-15)
    for t in range(steps):
        for k in SHAPES:
            g = torch.from_numpy(grads[t][k]).to(gpu)
            a[k].grad, b[k].grad = g.clone(), g.clone()
        mine.step(); ref.step()
    for k in SHAPES:
        d = (a[k] - b[k]).abs().max().item()
        assert d <= 4e-7 * max(1.0, LRS[k] / 1e-4), (k, d)
    mine.zero_grad()
    assert all(a[k].grad is None for k in SHAPES)
