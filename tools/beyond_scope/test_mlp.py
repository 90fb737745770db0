"""Split-K matrix-core weight / bias gradient of the deformation heads' Linear layers (saro-gs_amd/fused_mlp.py over
gsrast_linear_wgrad; beyond SURVEY 8f).  Reference = the same products in fp64 (torch on the CPU)."""
import numpy as np
import pytest
import torch
import torch.nn as nn


def test_convert_heads_keeps_parameters_and_keys():
    import fused_mlp
    head = nn.Sequential(nn.Linear(41, 128), nn.ReLU(), nn.Linear(128, 128), nn.ReLU(), nn.Linear(128, 7))
    keys, w0 = list(head.state_dict().keys()), head[0].weight
    fused_mlp.convert_heads(head)
    assert list(head.state_dict().keys()) == keys and head[0].weight is w0
    assert all(isinstance(head[i], fused_mlp.SplitKLinear) for i in (0, 2, 4))
    x = torch.randn(5, 41)
    head(x).sum().backward()                       # CPU tensors take nn.Linear's own path
    assert head[0].weight.grad is not None


@pytest.mark.gpu
@pytest.mark.parametrize("M,N1,N2", [(100003, 128, 128), (5000, 3, 128), (5000, 7, 128), (40000, 48, 128), (30000, 128, 41), (9000, 64, 128),
                                     (9000, 1, 64), (9000, 128, 32), (1, 5, 9), (7, 96, 100), (0, 8, 8)])
def test_wgrad_against_fp64(gpu, M, N1, N2):
    from diff_gaussian_rasterization_ch3 import _C
    g = torch.Generator().manual_seed(M + N1 * 7 + N2)
    G = torch.randn((M, N1), generator=g)
    X = torch.randn((M, N2), generator=g)
    want_w = (G.double().t() @ X.double()).numpy()
    want_b = G.double().sum(0).numpy()
    Gd, Xd = G.to(gpu), X.to(gpu)
    dW = torch.full((N1, N2), 7.0, device=gpu)
    db = torch.full((N1,), 7.0, device=gpu)
    rc = _C.lib().gsrast_linear_wgrad(M, N1, N2, Gd.data_ptr() if M else None, Xd.data_ptr() if M else None, dW.data_ptr(), db.data_ptr(), 0,
                                      torch.cuda.current_stream(gpu).cuda_stream)
    assert rc == 0
    tol = 2e-6 * max(1.0, np.sqrt(M)) * 4            # fp32 sums of M products of unit normals
    assert np.abs(dW.cpu().numpy() - want_w).max() <= tol
    assert np.abs(db.cpu().numpy() - want_b).max() <= tol
    # accumulate = 1 adds on top
    rc = _C.lib().gsrast_linear_wgrad(M, N1, N2, Gd.data_ptr() if M else None, Xd.data_ptr() if M else None, dW.data_ptr(), None, 1,
                                      torch.cuda.current_stream(gpu).cuda_stream)
    assert rc == 0
    assert np.abs(dW.cpu().numpy() - 2 * want_w).max() <= 2 * tol


@pytest.mark.gpu
def test_head_gradients_match_nn_linear(gpu):
    """A whole reference-shaped head (saro_gaussian.py:104-110) through SplitKLinear vs the same head in fp64."""
    import fused_mlp
    torch.manual_seed(3)
    P = 20011
    ref = nn.Sequential(nn.Linear(41, 128), nn.ReLU(), nn.Linear(128, 128), nn.ReLU(), nn.Linear(128, 48)).double()
    head = nn.Sequential(nn.Linear(41, 128), nn.ReLU(), nn.Linear(128, 128), nn.ReLU(), nn.Linear(128, 48))
    head.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    head = fused_mlp.convert_heads(head.to(gpu))
    x = torch.randn(P, 41)
    dy = torch.randn(P, 48)
    xr = x.double().requires_grad_(True)
    ref(xr).backward(dy.double())
    xg = x.to(gpu).requires_grad_(True)
    out = head(xg)
    out.backward(dy.to(gpu))
    assert float((out.detach().cpu().double() - ref(xr).detach()).abs().max()) < 1e-4
    assert float((xg.grad.cpu().double() - xr.grad).abs().max()) < 1e-4 * float(xr.grad.abs().max())
    for (k, p), (_, q) in zip(head.named_parameters(), ref.named_parameters()):
        err = float((p.grad.cpu().double() - q.grad).abs().max())
        assert err < 2e-5 * max(1.0, float(q.grad.abs().max())), (k, err)


@pytest.mark.gpu
def test_wgrad_rejects_wide_layers(gpu):
    from diff_gaussian_rasterization_ch3 import _C
    t = torch.zeros(8, device=gpu)
    assert _C.lib().gsrast_linear_wgrad(1, 129, 8, t.data_ptr(), t.data_ptr(), t.data_ptr(), None, 0, None) != 0
    import fused_mlp
    wide = fused_mlp.SplitKLinear(200, 16).to(gpu)          # falls back to nn.Linear's own backward
    wide(torch.randn(4, 200, device=gpu)).sum().backward()
    assert wide.weight.grad is not None
