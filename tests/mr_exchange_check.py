"""Helper of tests/test_gpu_multirank.py (run under torch.distributed.run, 2 ranks, gloo, both ranks on cuda:0):
one view per rank; the batch-mean leaf gradients from the factor exchange (all-reduce 11 + all-gather 3 floats per
Gaussian, recombined locally) must equal those from the plain all-reduce of all 59 floats."""
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
    worst = 0.0
    for k in res["allreduce"]:
        for other_mode in ("factors", "factors_overlapped"):
            a, b = res["allreduce"][k], res[other_mode][k]
            err = ((a - b).abs() / (1e-6 + 1e-4 * a.abs())).max().item()     # <= 1: within 1e-6 abs + 1e-4 rel
            worst = max(worst, err)
        assert a.abs().max().item() > 0, k
    # every rank must hold the same averaged gradient
    flat = torch.cat([v.reshape(-1) for v in list(res["factors"].values()) + list(res["factors_overlapped"].values())])
    other = flat.clone()
    torch.distributed.broadcast(other, src=0)
    same = bool(((flat - other).abs() <= 1e-7 + 1e-5 * other.abs()).all())
    print(f"EXCHANGE_CHECK rank {rank} worst {worst:.3f} same_on_all_ranks {same}", flush=True)
    if worst > 1.0 or not same:
        sys.exit(3)


if __name__ == "__main__":
    main()
