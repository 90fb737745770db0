"""One-view-per-GPU data parallelism for the rasterizer hot path.

The reference has no distributed code: train.py:198-226 renders the `opt.batch` views of an iteration
one after the other on one GPU and SUMS their gradients by hand
(scene/saro_gaussian.py:226-247 cache_gradient, :266-276 set_batch_gradient divides by the batch).
Each view's forward + backward only reads the (replicated) Gaussian attributes, so the views shard
with no data-path exchange; the one real exchange step is the gradient sum.  This module is that
step: one process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI on ROCm; "gloo" on
CPU for tests), rank r renders view r, then ONE flat fp32 buffer is all-reduced.

xGMI is point-to-point (7 links per GPU): a single large all-reduce over one flat buffer lets RCCL
pick its direct reduce-scatter + all-gather schedule across all links, instead of one latency-bound
collective per parameter tensor.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).
    Returns (rank, local_rank, world).  A single process (WORLD_SIZE unset or 1) needs no group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # GSRAST_DIST_BACKEND=gloo lets a 1-GPU box exercise the multi-rank code path (tests only)
            backend = os.environ.get("GSRAST_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def views_of_rank(n_views: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment of the iteration's views to ranks (batch == world: one view each)."""
    return list(range(rank, n_views, world))


class FlatGradBucket:
    """Packs the gradients of a fixed list of tensors into one contiguous fp32 buffer, all-reduces it
    (SUM) and scatters the mean back -- semantics of set_batch_gradient (saro_gaussian.py:269-276)."""

    def __init__(self, params: Sequence[torch.Tensor]):
        self.params = list(params)
        self.sizes = [p.numel() for p in self.params]
        total = sum(self.sizes)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views = []
        o = 0
        for p, n in zip(self.params, self.sizes):
            self.views.append(self.flat[o:o + n].view(p.shape))
            o += n

    def nbytes(self) -> int:
        return self.flat.numel() * 4

    def pack(self) -> None:
        for v, p in zip(self.views, self.params):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)

    def allreduce_mean(self, batch: int, async_op: bool = False):
        """SUM over ranks, then / batch.  batch = number of views in the iteration (== world when
        every rank renders one view)."""
        work = None
        if dist.is_initialized() and dist.get_world_size() > 1:
            work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)
        if work is None or not async_op:
            self.flat.mul_(1.0 / batch)
        return work

    def unpack(self) -> None:
        for v, p in zip(self.views, self.params):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)


_AVG_OK = True


def allreduce_mean_inplace(flat: torch.Tensor, batch: int) -> None:
    """Mean over the batch of one flat gradient buffer, in place.  RCCL's AVG does the division inside
    the collective (no extra pass over the buffer); gloo has no AVG, so SUM then scale."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        if batch != 1:
            flat.mul_(1.0 / batch)
        return
    global _AVG_OK
    if _AVG_OK and dist.get_backend() == "nccl" and batch == dist.get_world_size():
        try:
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)
            return
        except RuntimeError:            # a collective library without AVG: raised before anything is enqueued
            _AVG_OK = False
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.mul_(1.0 / batch)


def exchange_gradients(arena, means3D: torch.Tensor, batch: int) -> None:
    """Cross-rank mean of one view-per-rank backward whose leaf gradients live in a GradArena built with
    sh_factors=True (diff_gaussian_rasterization_ch3/_C.py).  Same result as all-reducing all 59 floats per Gaussian
    (up to fp32 summation order), with 2.6x fewer bytes on the links:

      * the dense part (means3D, opacity, scales, rotations: 11 floats / Gaussian) is all-reduced in place;
      * dL/dsh (48 floats / Gaussian) is NOT exchanged.  Each view's dL/dsh row k is w_k(view direction) * g with g the
        view's clamp-masked colour gradient (3 floats): ranks all-gather g (+ their camera position, 3 floats) and every
        rank evaluates  (1/batch) * sum_r w(dir_r) (x) g_r  itself (gsrast_sh_grad_combine, one HIP kernel).

    xGMI is point-to-point, 7 links per GPU: an all-gather of 12 B/Gaussian/rank plus an all-reduce of 44 B/Gaussian
    moves ~160 B per Gaussian and rank at 8 GPUs, the plain all-reduce of 236 B/Gaussian moves ~410 B."""
    from diff_gaussian_rasterization_ch3 import _C
    if not getattr(arena, "sh_factors", False):
        raise ValueError("exchange_gradients needs GradArena(..., sh_factors=True)")
    multi = dist.is_initialized() and dist.get_world_size() > 1
    n_views = dist.get_world_size() if multi else 1
    if n_views != arena.world:
        raise ValueError(f"arena was built for {arena.world} ranks, the process group has {n_views}")
    allreduce_mean_inplace(arena.dense, batch)
    if multi:
        dist.all_gather_into_tensor(arena.gathered, arena.factor)
        chunks = arena.gathered
    else:
        chunks = arena.factor
    _C.sh_grad_combine(arena, means3D, chunks, n_views, 1.0 / batch)


def reduce_densification_stats(point_grad_norm: torch.Tensor, visible_count: torch.Tensor,
                               max_radii: torch.Tensor) -> None:
    """In-place cross-rank reduction of the densification statistics of train.py:282-292:
    SUM of the screen-space gradient norms and visibility counts, MAX of the radii."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(point_grad_norm, op=dist.ReduceOp.SUM)
        dist.all_reduce(visible_count, op=dist.ReduceOp.SUM)
        dist.all_reduce(max_radii, op=dist.ReduceOp.MAX)


def max_over_ranks(x: float, device: torch.device) -> float:
    if dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([x], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return float(x)


def barrier() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
