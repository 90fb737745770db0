"""Helper of tests/test_gpu_multirank.py (run under torch.distributed.run, 2 ranks, gloo, both ranks on cuda:0):
one view per rank; the batch-mean leaf gradients from the factor exchange (all-reduce 11 + all-gather 3 floats per
Gaussian, recombined locally) must equal those from the plain all-reduce of all 59 floats -- also with only the touched rows
travelling (sparse), and for the RAW leaves of GaussianRasterizerRaw (raw GradArena)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "saro-gs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch  # noqa: E402
import bench  # noqa: E402
import scenes  # noqa: E402
import view_parallel as vp  # noqa: E402
import diff_gaussian_rasterization_ch3 as rast  # noqa: E402
from diff_gaussian_rasterization_ch3 import _C  # noqa: E402


def main():
    rank, local, world = vp.init_from_env()
    dev = torch.device("cuda:0")
    P, W, H, deg = 20000, 320, 240, 3
    wl = bench.Workload(rast, scenes, P, W, H, deg, view_k=rank, n_views=world, dev=dev)
    res = {}
    started = []
    for mode in ("allreduce", "factors", "factors_overlapped"):
        arena = _C.GradArena(P, 16, dev, sh_factors=(mode != "allreduce"), world=world)
        _C.set_grad_arena(arena)
        if mode == "factors_overlapped":      # the all-gather starts inside the backward, between its two phases
            vp.overlap_factor_exchange(True)
            inner = _C._factor_ready_hook
            _C.set_factor_ready_hook(lambda ar: (inner(ar), started.append(getattr(ar, "_gather_work", None) is not None)))
        wl.step(arena, world)
        torch.cuda.synchronize()
        res[mode] = {k: v.grad.detach().clone() for k, v in wl.leaves.items()}
        _C.set_grad_arena(None)
        vp.overlap_factor_exchange(False)
    assert started == [True] and getattr(arena, "_gather_work", None) is None, started
    # only the rows some rank touched travel (exchange_gradients(sparse=True)): same mean
    arena = _C.GradArena(P, 16, dev, sh_factors=True, world=world)
    _C.set_grad_arena(arena)
    wl.sparse = True
    wl.step(arena, world)
    torch.cuda.synchronize()
    res["factors_sparse"] = {k: v.grad.detach().clone() for k, v in wl.leaves.items()}
    sent = dict(wl.exchanged)
    _C.set_grad_arena(None)
    wl.sparse = False
    assert 0 < sent["rows"] < P and sent["allreduce"] < P * 44, sent          # (fewer rows than Gaussians: most are never blended)

    # consecutive sparse steps over DIFFERENT views on ONE arena (round 5: the recombination writes the union's rows only and the
    # exchange clears the previous union's -- a row that leaves the union must read zero again), against the plain all-reduce
    # ... and the all-gather exchange (sparse="gather"), alternating with the other forms on the same arena: each form keeps its own record
    # of the SH rows it wrote, a change of form clears everything once
    wa = bench.Workload(rast, scenes, P, W, H, deg, view_k=rank, n_views=5, dev=dev, poses=3)
    wb = bench.Workload(rast, scenes, P, W, H, deg, view_k=rank, n_views=5, dev=dev, poses=3)
    arena_a, arena_b = _C.GradArena(P, 16, dev, world=world), _C.GradArena(P, 16, dev, sh_factors=True, world=world)
    worst_seq, unions = 0.0, []
    forms = [True, True, True, False, True, "gather", "gather", "gather", True, "gather", False, "gather"]
    for form in forms:                  # (False = the dense combine: it voids what is known about the rows, the next sparse step clears all)
        wb.sparse = form
        _C.set_grad_arena(arena_a)
        wa.step(arena_a, world)
        _C.set_grad_arena(arena_b)
        wb.step(arena_b, world)
        torch.cuda.synchronize()
        unions.append(wb.exchanged["rows"])
        for k in wa.leaves:
            a, b = wa.leaves[k].grad, wb.leaves[k].grad
            err = ((a - b).abs() / (1e-6 + 1e-4 * a.abs())).max().item()
            assert err <= 1.0, (form, len(unions), k, err)
            worst_seq = max(worst_seq, err)
    _C.set_grad_arena(None)
    assert worst_seq <= 1.0 and len(set(unions[:3])) > 1, (worst_seq, unions)     # (the unions differed from step to step)
    res["factors_gather"] = {k: v.grad.detach().clone() for k, v in wb.leaves.items()}      # (the last step: must be the same on every rank, bit for bit)

    # the RAW leaves (GaussianRasterizerRaw: SaRO-GS's call pattern, `shs` is cat(features_dc, features_rest), never a leaf)
    L = wl.leaves
    raw = dict(xyz=L["means3D"].detach().clone(), rotation=L["rotations"].detach().clone(), scaling=torch.log(L["scales"].detach()),
               opacity=torch.logit(L["opacities"].detach().clamp(1e-4, 1 - 1e-4)), f_dc=L["shs"].detach()[:, :1].contiguous(),
               f_rest=L["shs"].detach()[:, 1:].contiguous())
    raw = {k: v.requires_grad_(True) for k, v in raw.items()}
    raster_raw = rast.GaussianRasterizerRaw(wl.rs)
    m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
    res_raw = {}
    for mode in ("allreduce", "factors", "factors_sparse", "factors_gather"):
        arena = _C.GradArena(P, 16, dev, sh_factors=(mode != "allreduce"), world=world, raw=True)
        _C.set_grad_arena(arena)
        for v in list(raw.values()) + [m2]:
            v.grad = None
        arena.zero_grad()
        color, _, _ = raster_raw(raw["xyz"], m2, raw["rotation"], raw["scaling"], raw["opacity"], raw["f_dc"], raw["f_rest"])
        color.backward(wl.g)
        if mode == "allreduce":
            vp.allreduce_mean_inplace(arena.flat, world)
        else:
            vp.exchange_gradients(arena, raw["xyz"].detach(), world, sparse=("gather" if mode == "factors_gather" else mode == "factors_sparse"))
        torch.cuda.synchronize()
        assert all(v.grad.data_ptr() >= arena.flat.data_ptr() and v.grad.data_ptr() < arena.flat.data_ptr() + arena.flat.numel() * 4 for v in raw.values()), "the raw leaves' gradients must be views of the bucket"
        res_raw[mode] = {k: v.grad.detach().clone() for k, v in raw.items()}
        _C.set_grad_arena(None)
    worst = 0.0
    for k in res_raw["allreduce"]:
        for other_mode in ("factors", "factors_sparse", "factors_gather"):
            a, b = res_raw["allreduce"][k], res_raw[other_mode][k]
            worst = max(worst, ((a - b).abs() / (1e-6 + 1e-4 * a.abs())).max().item())
        assert res_raw["allreduce"][k].abs().max().item() > 0, k
    for k in res["allreduce"]:
        for other_mode in ("factors", "factors_overlapped", "factors_sparse"):
            a, b = res["allreduce"][k], res[other_mode][k]
            err = ((a - b).abs() / (1e-6 + 1e-4 * a.abs())).max().item()     # <= 1: within 1e-6 abs + 1e-4 rel
            worst = max(worst, err)
        assert a.abs().max().item() > 0, k
    # every rank must hold the same averaged gradient
    flat = torch.cat([v.reshape(-1) for v in list(res["factors"].values()) + list(res["factors_overlapped"].values()) + list(res["factors_sparse"].values())
                      + list(res_raw["factors"].values()) + list(res_raw["factors_sparse"].values())])
    exact = torch.cat([v.reshape(-1) for v in list(res["factors_gather"].values()) + list(res_raw["factors_gather"].values())])
    exact_other = exact.clone()
    torch.distributed.broadcast(exact_other, src=0)
    assert torch.equal(exact, exact_other), "the all-gather exchange adds the chunks in rank order: every rank must hold the same bits"
    other = flat.clone()
    torch.distributed.broadcast(other, src=0)
    same = bool(((flat - other).abs() <= 1e-7 + 1e-5 * other.abs()).all())
    avg_ok = getattr(vp, "_AVG_OK", None)        # (False: the collective library rejected ReduceOp.AVG and SUM + scale took over)
    print(f"EXCHANGE_CHECK rank {rank} worst {worst:.3f} same_on_all_ranks {same} backend {torch.distributed.get_backend()} world {world} avg_ok {avg_ok}", flush=True)
    if worst > 1.0 or not same:
        sys.exit(3)


if __name__ == "__main__":
    main()
