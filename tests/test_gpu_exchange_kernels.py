"""-m gpu: the kernels of the multi-GPU gradient exchange one by one against torch (include/gsrast.h: gsrast_rows_pack / _unpack,
gsrast_grad_rows_pack / _clear / _add, gsrast_sh_grad_combine_union) -- the exchange tests (mr_exchange_check.py, test_view_parallel.py)
cover them end to end; here every row, ragged sizes, both arenas."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sh_weights(d, deg):
    """w_k(direction) of forward.cu:20-71 / backward.cu:78-141 in fp64: [n, 16]."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    C0, C1 = 0.28209479177387814, 0.4886025119029199
    C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
    C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658, 1.445305721320277, -0.5900435899266435)
    w = torch.zeros((d.shape[0], 16), dtype=d.dtype, device=d.device)
    w[:, 0] = C0
    if deg > 0:
        w[:, 1], w[:, 2], w[:, 3] = -C1 * y, C1 * z, -C1 * x
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        w[:, 4], w[:, 5], w[:, 6], w[:, 7], w[:, 8] = C2[0] * xy, C2[1] * yz, C2[2] * (2 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)
    if deg > 2:
        w[:, 9], w[:, 10] = C3[0] * y * (3 * xx - yy), C3[1] * xy * z
        w[:, 11], w[:, 12] = C3[2] * y * (4 * zz - xx - yy), C3[3] * z * (2 * zz - 3 * xx - 3 * yy)
        w[:, 13], w[:, 14], w[:, 15] = C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy), C3[6] * x * (xx - 3 * yy)
    return w


@pytest.mark.parametrize("P,n", [(1000, 137), (70000, 33333), (513, 513), (300, 0)])
def test_rows_pack_and_unpack(P, n, rast, gpu):
    _C = rast._C
    g = torch.Generator(device="cpu").manual_seed(P + n)
    arrays = [torch.randn((P, w), generator=g).to(gpu) for w in (3, 1, 3, 4)]
    idx = torch.randperm(P, generator=g)[:n].sort().values.to(gpu)
    packed = torch.full((n, 11), 7.0, device=gpu)
    _C.rows_pack(idx, arrays, packed)
    assert torch.equal(packed, torch.cat([a[idx] for a in arrays], 1))
    before = [a.clone() for a in arrays]
    packed.mul_(2.0)
    _C.rows_pack(idx, arrays, packed, unpack=True)
    o = 0
    for a, b in zip(arrays, before):
        want = b.clone()
        want[idx] = packed[:, o: o + a.shape[1]]
        o += a.shape[1]
        assert torch.equal(a, want)                    # (rows outside idx untouched)


@pytest.mark.parametrize("raw", [False, True], ids=["rasterizer_leaves", "raw_leaves"])
@pytest.mark.parametrize("P,deg,M,frac", [(5000, 3, 16, 0.1), (70001, 2, 16, 0.5), (4097, 1, 4, 1.0), (900, 3, 16, 0.0)])
def test_grad_rows_pack_clear_add(P, deg, M, frac, raw, rast, gpu):
    """One rank's touched rows through pack -> (what an all-gather would hand back) -> clear -> add, against torch in fp64."""
    _C = rast._C
    g = torch.Generator(device="cpu").manual_seed(P * 7 + M)
    arena = _C.GradArena(P, M, gpu, sh_factors=True, world=1, raw=raw)
    arena.last_degree = deg
    touched = (torch.rand(P, generator=g) < frac).to(torch.uint8).to(gpu)
    segs = arena.dense_segments()
    for sg in segs:
        sg.copy_(torch.randn(sg.shape, generator=g))
    arena.factor[: 3 * P] = torch.randn(3 * P, generator=g).to(gpu)
    campos = torch.tensor([0.3, -2.0, 4.5], device=gpu)
    arena.factor[3 * P: 3 * P + 3] = campos
    means = (torch.randn((P, 3), generator=g) * 2).to(gpu)
    idx = torch.nonzero(touched).squeeze(1)
    n = int(idx.numel())
    rows = torch.full((P + 1, 16), -1, dtype=torch.int32, device=gpu)
    rows[0].zero_()
    rows[0, 1:4] = campos.view(torch.int32)
    _C.grad_rows_pack(arena, touched, rows)
    torch.cuda.synchronize()
    assert int(rows[0, 0]) == n
    body = rows[1: 1 + n]
    order = torch.argsort(body[:, 0])                  # (rows arrive in any order)
    body = body[order]
    assert torch.equal(body[:, 0].long(), idx)
    dense0 = torch.cat(segs, 1).clone()
    assert torch.equal(body[:, 1:12].contiguous().view(torch.float32), dense0[idx])
    assert torch.equal(body[:, 12:15].contiguous().view(torch.float32), arena.factor[: 3 * P].view(P, 3)[idx])
    assert bool((body[:, 15] == 0).all()) and bool((rows[1 + n:] == -1).all())      # (nothing written past the count)
    # clear: the dense rows of the list go to zero, the SH region's too; other rows keep what they held
    sh_views = [v.view(P, -1) for v in _C._arena_sh_arrays(arena) if v is not None]
    for v in sh_views:
        v.fill_(3.0)
    chunk = rows[: 1 + n].contiguous()
    _C.grad_rows_clear(arena, chunk.unsqueeze(0), dense=True, sh=True)
    keep = touched == 0
    for sg, d0 in zip(segs, torch.split(dense0, [3, 1, 3, 4], 1)):
        assert bool((sg[idx] == 0).all()) and torch.equal(sg[keep], d0[keep])
    for v in sh_views:
        assert bool((v[idx] == 0).all()) and bool((v[keep] == 3.0).all())
        v.zero_()
    # add twice (two "ranks" with the same rows): dense = 2 * scale * row, dL/dsh = 2 * scale * w(dir) (x) factor
    scale = 0.5
    _C.grad_rows_add(arena, chunk, means, scale)
    _C.grad_rows_add(arena, chunk, means, scale)
    torch.cuda.synchronize()
    for sg, d0 in zip(segs, torch.split(dense0, [3, 1, 3, 4], 1)):
        assert torch.allclose(sg[idx], 2 * scale * d0[idx], rtol=1e-6, atol=1e-7)
        assert torch.equal(sg[keep], d0[keep])
    d = (means[idx] - campos).double()
    d = d / d.norm(dim=1, keepdim=True)
    fac = arena.factor[: 3 * P].view(P, 3)[idx].double()
    want = (_sh_weights(d, deg)[:, :M, None] * fac[:, None, :] * (2 * scale)).reshape(n, M * 3)
    got = torch.cat([v[idx] for v in sh_views], 1).double() if raw else sh_views[0][idx].double()
    if n:
        err = ((got - want).abs() / (1e-6 + 1e-5 * want.abs())).max().item()
        assert err <= 1.0, err
    for v in sh_views:
        assert bool((v[keep] == 0).all())


@pytest.mark.parametrize("M,deg", [(16, 3), (4, 1)])
def test_sh_grad_combine_union_writes_the_union_only(M, deg, rast, gpu):
    _C = rast._C
    P, n_views = 20000, 3
    g = torch.Generator(device="cpu").manual_seed(M)
    arena = _C.GradArena(P, M, gpu, sh_factors=True, world=n_views)
    arena.last_degree = deg
    means = (torch.randn((P, 3), generator=g) * 2).to(gpu)
    idx = torch.randperm(P, generator=g)[:4321].sort().values.to(gpu)
    n = int(idx.numel())
    stride = ((3 * n + 3 + 3) // 4) * 4
    chunks = torch.zeros(n_views * stride, device=gpu)
    cams = torch.randn((n_views, 3), generator=g).to(gpu) * 5
    facs = torch.randn((n_views, n, 3), generator=g).to(gpu)
    for r in range(n_views):
        chunks[r * stride: r * stride + 3 * n] = facs[r].reshape(-1)
        chunks[r * stride + 3 * n: r * stride + 3 * n + 3] = cams[r]
    sh = _C.sh_grad_combine(arena, means, chunks, n_views, 0.25, chunk_stride=stride, idx=idx)
    want = torch.zeros((n, M, 3), dtype=torch.float64, device=gpu)
    for r in range(n_views):
        d = (means[idx] - cams[r]).double()
        d = d / d.norm(dim=1, keepdim=True)
        want += _sh_weights(d, deg)[:, :M, None] * facs[r].double()[:, None, :]
    want *= 0.25
    err = ((sh[idx].double() - want).abs() / (1e-6 + 1e-5 * want.abs())).max().item()
    assert err <= 1.0, err
    keep = torch.ones(P, dtype=torch.bool, device=gpu)
    keep[idx] = False
    assert bool((sh[keep] == 0).all())
    # a second union: the first one's rows read zero again unless they are in it
    idx2 = torch.randperm(P, generator=g)[:999].sort().values.to(gpu)
    stride2 = ((3 * 999 + 3 + 3) // 4) * 4
    chunks2 = torch.randn(n_views * stride2, generator=g).to(gpu)
    sh2 = _C.sh_grad_combine(arena, means, chunks2, n_views, 0.25, chunk_stride=stride2, idx=idx2)
    keep2 = torch.ones(P, dtype=torch.bool, device=gpu)
    keep2[idx2] = False
    assert bool((sh2[keep2] == 0).all()) and bool((sh2[idx2] != 0).any())


def test_grad_rows_clear_and_add_skip_indices_past_the_arrays(rast, gpu):
    """ABI 5: the indices inside a chunk come from a peer.  A row whose index is >= P -- a damaged or stale chunk -- is skipped by
    gsrast_grad_rows_clear and gsrast_grad_rows_add; the rows in range are processed as ever and nothing outside the arena changes."""
    _C = rast._C
    P, M, deg = 3000, 16, 3
    arena = _C.GradArena(P, M, gpu, sh_factors=True, world=1)
    arena.last_degree = deg
    g = torch.Generator(device="cpu").manual_seed(5)
    for sg in arena.dense_segments():
        sg.copy_(torch.randn(sg.shape, generator=g))
    sh = _C._arena_sh_arrays(arena)[0].view(P, -1)
    sh.fill_(1.0)
    means = torch.randn((P, 3), generator=g).to(gpu)
    n = 8
    chunk = torch.zeros((1 + n, 16), dtype=torch.int32, device=gpu)
    chunk[0, 0] = n
    chunk[0, 1:4] = torch.tensor([0.0, 0.0, 5.0], device=gpu).view(torch.int32)
    idx = torch.tensor([5, P, 17, P + 12345, 2_000_000_000, P - 1, 0x7FFFFFFF, 44], dtype=torch.int32)      # four good rows, four far outside
    chunk[1:, 0] = idx.to(gpu)
    chunk[1:, 1:15] = torch.ones((n, 14), device=gpu).view(torch.int32)
    guard = torch.full((1 << 20,), 7.0, device=gpu)            # (something allocated behind the arena: must keep its contents)
    dense0 = [sg.clone() for sg in arena.dense_segments()]
    good = torch.tensor([5, 17, P - 1, 44], device=gpu)
    _C.grad_rows_add(arena, chunk, means, 1.0)
    torch.cuda.synchronize()
    for sg, d0 in zip(arena.dense_segments(), dense0):
        want = d0.clone()
        want[good] += 1.0
        assert torch.allclose(sg, want)
    changed = (sh != 1.0).any(dim=1)
    assert set(torch.nonzero(changed).flatten().tolist()) == {5, 17, 44, P - 1}
    _C.grad_rows_clear(arena, chunk.unsqueeze(0), dense=True, sh=True)
    torch.cuda.synchronize()
    for sg, d0 in zip(arena.dense_segments(), dense0):
        keep = torch.ones(P, dtype=torch.bool, device=gpu)
        keep[good] = False
        assert bool((sg[good] == 0).all()) and torch.equal(sg[keep], d0[keep])
    assert bool((sh[good] == 0).all()) and bool((guard == 7.0).all())
