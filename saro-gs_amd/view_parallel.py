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

import contextlib
import torch
import torch.distributed as dist


_FORCE_COLLECTIVES = os.environ.get("GSRAST_FORCE_COLLECTIVES", "") == "1"
# GSRAST_GATHER_ASYNC_CAP=0: the all-gather exchange agrees on its row capacity with a synchronous 4-byte all-reduce in every step instead of
# beside the backward (_touched_hook) -- one host synchronisation per step more, no collective issued from inside the backward
_ASYNC_CAPACITY = os.environ.get("GSRAST_GATHER_ASYNC_CAP", "1") != "0"


def force_collectives(on: bool = True) -> None:
    """Run every collective of this module even in a process group of ONE rank (default: a single rank short-circuits them).
    A 1-GPU box can then drive the exchange through RCCL itself -- `backend="nccl"`, world size 1: ReduceOp.AVG, the uint8 MAX of the
    sparse exchange, all_gather_into_tensor, the asynchronous all-reduce of distributed_step -- which is the library that runs them
    on the 8-GPU node (tests/test_gpu_rccl.py, `bench.py --gpus 1 --force-collectives`).  Also: GSRAST_FORCE_COLLECTIVES=1."""
    global _FORCE_COLLECTIVES
    _FORCE_COLLECTIVES = bool(on)


def collectives_active() -> bool:
    """Is there a process group whose collectives this module should call (more than one rank, or forced)?"""
    return dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE_COLLECTIVES)


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).
    Returns (rank, local_rank, world).  A single process (WORLD_SIZE unset or 1) needs no group -- unless force_collectives()."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or _FORCE_COLLECTIVES) and not dist.is_initialized():
        if backend is None:
            # GSRAST_DIST_BACKEND=gloo lets a 1-GPU box exercise the multi-rank code path (tests only)
            backend = os.environ.get("GSRAST_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


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
        if collectives_active():
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
    if not collectives_active():
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


def exchange_gradients(arena, means3D: torch.Tensor, batch: int, sparse=False) -> Dict[str, int]:
    """Cross-rank mean of one view-per-rank backward whose leaf gradients live in a GradArena built with
    sh_factors=True (diff_gaussian_rasterization_ch3/_C.py; raw=True for GaussianRasterizerRaw's six leaves -- SaRO-GS's own call
    pattern, where `shs` is never a leaf).  Same result as all-reducing all 59 floats per Gaussian (up to fp32 summation order),
    with 2.6x fewer bytes on the links:

      * the dense part (means3D, opacity, scales, rotations: 11 floats / Gaussian) is all-reduced in place;
      * dL/dsh (48 floats / Gaussian) is NOT exchanged.  Each view's dL/dsh row k is w_k(view direction) * g with g the
        view's clamp-masked colour gradient (3 floats): ranks all-gather g (+ their camera position, 3 floats) and every
        rank evaluates  (1/batch) * sum_r w(dir_r) (x) g_r  itself (gsrast_sh_grad_combine_rows, one HIP kernel).

    xGMI is point-to-point, 7 links per GPU: an all-gather of 12 B/Gaussian/rank plus an all-reduce of 44 B/Gaussian
    moves ~160 B per Gaussian and rank at 8 GPUs, the plain all-reduce of 236 B/Gaussian moves ~410 B.

    sparse=True (round 4): only the rows SOME rank touched travel.  A view's gradient rows are exactly zero for every Gaussian it
    did not blend -- culled, left out by the list cut, or occluded: ~90 % of the 3 M bench scene per view -- so ranks first take the
    MAX of a one-byte "my rows are non-zero" flag per Gaussian (P bytes all-reduced), then exchange the union's rows only: a
    compacted [n, 11] all-reduce and a compacted [n, 3] all-gather; rows outside the union are zero on every rank and stay so.
    One host synchronisation per step (the union's size), beside the one every forward already has.

    sparse="gather" (round 5): every rank sends only the rows its OWN view touched, in one all-gather -- see _exchange_gather.

    Returns the bytes this rank handed to the collectives: {"allreduce", "allgather", "rows"}."""
    from diff_gaussian_rasterization_ch3 import _C
    if not getattr(arena, "sh_factors", False):
        raise ValueError("exchange_gradients needs GradArena(..., sh_factors=True)")
    multi = collectives_active()
    n_views = dist.get_world_size() if multi else 1
    if n_views != arena.world:
        raise ValueError(f"arena was built for {arena.world} ranks, the process group has {n_views}")
    P = arena.P
    pending = getattr(arena, "_gather_work", None)
    if sparse and pending is not None:
        # (the asynchronous gather moves EVERY row's factor: the two forms do not combine -- say so instead of silently paying for both)
        import warnings
        warnings.warn("exchange_gradients(sparse=True) with overlap_factor_exchange() in force: the dense factor gather has already "
                      "been started inside the backward, the exchange falls back to the dense form; switch the overlap off "
                      "(overlap_factor_exchange(False)) to send only the touched rows", RuntimeWarning, stacklevel=2)
    if sparse == "gather" and (arena.M * 3) % 4 != 0:
        sparse = True                   # (the row kernels move SH rows in 16-byte pieces: M = 4, 8, 12, 16; other M take the union form)
    if sparse == "gather" and multi and pending is None and P > 0:
        return _exchange_gather(arena, means3D, batch, n_views)
    _disarm_gather(arena)           # (another form was asked for: no capacity collective inside later backwards, no side-stream reader of the flags)
    if sparse and multi and pending is None and P > 0:
        segs = arena.dense_segments()
        fac = arena.factor[: 3 * P].view(P, 3)
        touched = _touched_flags(arena, segs, fac)
        dist.all_reduce(touched, op=dist.ReduceOp.MAX)
        idx = torch.nonzero(touched, as_tuple=False).squeeze(1)          # (host synchronisation: every rank learns the same n)
        n = int(idx.numel())
        stride = ((3 * n + 3 + 3) // 4) * 4
        mine = torch.empty(stride, dtype=torch.float32, device=fac.device)
        if fac.is_cuda:                 # one launch each way (gsrast_rows_pack) instead of an index_select / index_copy_ per array
            comp = _C.rows_pack(idx, segs, torch.empty((n, sum(sg.shape[1] for sg in segs)), dtype=torch.float32, device=fac.device))
            allreduce_mean_inplace(comp.view(-1), batch)
            _C.rows_pack(idx, segs, comp, unpack=True)
            _C.rows_pack(idx, [fac], mine[: 3 * n].view(n, 3))
        else:                           # (CPU tensors: the gloo tests of this function)
            comp = torch.cat([sg.index_select(0, idx) for sg in segs], dim=1).contiguous()       # [n, 11]
            allreduce_mean_inplace(comp.view(-1), batch)
            o = 0
            for sg in segs:
                sg.index_copy_(0, idx, comp[:, o: o + sg.shape[1]])
                o += sg.shape[1]
            mine[: 3 * n] = fac.index_select(0, idx).reshape(-1)
        mine[3 * n:] = 0.0
        mine[3 * n: 3 * n + 3] = arena.factor[3 * P: 3 * P + 3]
        gathered = torch.empty(n_views * stride, dtype=torch.float32, device=fac.device)
        dist.all_gather_into_tensor(gathered, mine)
        # round 5: the recombination writes the union's rows only (29 MB of 576 at 3 M); rows outside it are kept zero from step to step
        _C.sh_grad_combine(arena, means3D, gathered, n_views, 1.0 / batch, chunk_stride=stride, idx=idx)
        return {"allreduce": comp.numel() * 4 + P, "allgather": stride * 4, "rows": n}
    allreduce_mean_inplace(arena.dense, batch)
    if multi:
        if pending is not None:                    # started inside the backward (overlap_factor_exchange)
            pending.wait()
            arena._gather_work = None
        else:
            dist.all_gather_into_tensor(arena.gathered, arena.factor)
        chunks = arena.gathered
    else:
        chunks = arena.factor
    _C.sh_grad_combine(arena, means3D, chunks, n_views, 1.0 / batch)
    return {"allreduce": arena.dense.numel() * 4 if multi else 0, "allgather": arena.chunk * 4 if multi else 0, "rows": P}


def _touched_flags(arena, segs, fac) -> torch.Tensor:
    """uint8 [P]: 1 = this rank's gradient row of the Gaussian may be non-zero."""
    P = arena.P
    if getattr(arena, "touched_fresh", False) and arena.touched.numel() == P:
        # round 5: the backward left one byte per Gaussian, "some pixel of this view consumed it" (the forward blend's untouched bits,
        # gsrast_touched_rows): a superset of the rows with a non-zero gradient, for 1 B instead of 56 B read per Gaussian
        arena.touched_fresh = False
        return arena.touched
    # any of the row's 14 floats is non-zero (all five arrays are looked at: a sum can cancel to exactly zero in one of them); the flag
    # lives on the arena, not in a fresh [P] tensor every step
    touched = getattr(arena, "_touched", None)
    if touched is None or touched.numel() != P:
        touched = arena._touched = torch.empty(P, dtype=torch.uint8, device=fac.device)
    torch.any(fac != 0, dim=1, out=touched.view(torch.bool))
    for sg in segs:
        touched |= (sg != 0).any(dim=1).view(torch.uint8)
    return touched


def _disarm_gather(arena) -> None:
    """The arena stops agreeing on the gather exchange's capacity inside its backwards (_touched_hook).  Called by every OTHER form of
    exchange_gradients -- on every rank alike, since every rank calls the same form -- so that a caller who switches forms does not keep
    issuing one collective per backward for good; a capacity event of the step just run is drained (its all-reduce was issued by every
    rank's backward: nothing is left half-done) before anybody overwrites the flags it reads on the side stream."""
    if getattr(arena, "_gather_armed", False):
        arena._gather_armed = False
    ev = getattr(arena, "_cap_event", None)
    if ev is not None:
        ev.synchronize()
        arena._cap_event = None
        arena.touched_reader_event = None


def _touched_hook(arena) -> None:
    """Inside the backward of an arena that exchanges by all-gather, before its kernels are enqueued: the ranks take the MAX of their
    touched-row counts on a side stream and the result travels to pinned host memory, all of it beside the backward -- the exchange
    then sizes its buffers without waiting for the device (the step's host synchronisation moves off the critical path).

    Contract (round 6): ONE collective per backward of an ARMED arena, on every rank -- the arena is armed by its first gather
    exchange and disarmed by any other form (_disarm_gather), so ranks that call the same sequence of backwards and exchanges issue the
    same sequence of collectives.  The event carries the sequence number of the backward that produced it (arena.touched_seq, counted by
    _C._export_touched); _exchange_gather consumes it only if it belongs to the LAST backward and falls back to the synchronous
    capacity all-reduce otherwise (two backwards in a step, an exchange called twice: decided alike on every rank)."""
    if not getattr(arena, "_gather_armed", False) or not collectives_active() or not arena.touched.is_cuda:
        return
    dev = arena.touched.device
    side = getattr(arena, "_cap_stream", None)
    if side is None:
        side = arena._cap_stream = torch.cuda.Stream(device=dev)
        arena._cap_host = torch.zeros(1, dtype=torch.int32).pin_memory()
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        cnt = arena.touched.sum(dtype=torch.int32).view(1)
        dist.all_reduce(cnt, op=dist.ReduceOp.MAX)
        arena._cap_host.copy_(cnt, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(side)
    arena._cap_event = arena.touched_reader_event = ev      # (_C._export_touched waits for it before it overwrites the flags: a step that never exchanged)
    arena._cap_seq = getattr(arena, "touched_seq", 0)        # the backward this capacity belongs to


def _exchange_gather(arena, means3D: torch.Tensor, batch: int, n_views: int) -> Dict[str, int]:
    """exchange_gradients(sparse="gather") (round 5; csrc/gsrast_exchange.h): every rank sends the rows ITS view touched -- 64 bytes each:
    index, the 11 dense floats, the 3 floats of the dL/dsh factor -- in ONE all-gather of chunks [header | cap rows], cap = the largest
    count of the step (a 4-byte MAX all-reduce, the step's one host synchronisation), and adds the chunks into its arrays in RANK order:
    no atomics, the same sums in the same order on every rank -- bit-identical gradients on all replicas.  3 M Gaussians, 8 ranks,
    ~150 k rows per view: 67 MB received per rank in one collective (sparse=True: 75 MB in three, plus two compaction passes).  From the
    second step on the capacity is agreed on beside the backward (_touched_hook): the exchange does not wait for the device."""
    from diff_gaussian_rasterization_ch3 import _C
    P = arena.P
    dev = arena.flat.device
    segs = arena.dense_segments()
    fac = arena.factor[: 3 * P].view(P, 3)
    touched = _touched_flags(arena, segs, fac)
    _check_gather_overflow(arena)       # (the PREVIOUS step's header counts against its capacity: a device flag read without a wait of its own)
    send = getattr(arena, "_rows_send", None)
    if send is None or send.shape[0] != P + 1 or send.device != dev:
        # (sized for every row: 64 B per Gaussian of address space, of which a step touches the header and its own rows; zeroed ONCE, so
        # that the rows between a step's count and the capacity that travel with the chunk are never uninitialised memory)
        send = arena._rows_send = torch.zeros((P + 1, _C.GRAD_ROW_WORDS), dtype=torch.int32, device=dev)
    send[0].zero_()
    send[0, 1:4] = arena.factor[3 * P: 3 * P + 3].view(torch.int32)
    _C.grad_rows_pack(arena, touched, send)
    ev = getattr(arena, "_cap_event", None)
    if ev is not None and getattr(arena, "_cap_seq", -1) != getattr(arena, "touched_seq", 0):
        ev.synchronize()                # a capacity agreed on for ANOTHER backward than the last one (two backwards in this step): not this step's
        ev, arena._cap_event, arena.touched_reader_event = None, None, None
    if ev is not None:
        # the ranks agreed on the capacity BESIDE the backward (_touched_hook): the host has had the number for a millisecond
        ev.synchronize()
        cap, arena._cap_event, arena.touched_reader_event = int(arena._cap_host[0]), None, None
    else:
        cmax = send[0, :1].clone()
        dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
        cap = int(cmax.item())                                         # (host synchronisation: every rank learns the same cap)
    if not getattr(arena, "_gather_armed", False) and _ASYNC_CAPACITY and not getattr(arena, "_gather_no_async", False):     # from the next backward on: the capacity is agreed on early
        arena._gather_armed = True
        _C.set_touched_ready_hook(_touched_hook)
    mine = send[: 1 + cap]
    gathered = torch.empty((n_views, 1 + cap, _C.GRAD_ROW_WORDS), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(gathered.view(-1), mine.reshape(-1))
    # the SH region is kept zero outside the rows the previous step wrote (the backward does not write it in factor mode); the dense
    # arrays hold this rank's own gradient: its rows go to zero and come back with everybody's, in rank order
    prev = getattr(arena, "_rows_prev", None)
    if getattr(arena, "sh_rows_known", False) and prev is not None:
        _C.grad_rows_clear(arena, prev, dense=False, sh=True)
    else:
        for v in _C._arena_sh_arrays(arena):
            if v is not None:
                v.zero_()
    _C.grad_rows_clear(arena, mine.unsqueeze(0), dense=True, sh=False)
    for r in range(n_views):
        _C.grad_rows_add(arena, gathered[r], means3D, 1.0 / batch)
    arena._rows_prev, arena._sh_union, arena.sh_rows_known = gathered, None, True
    # overflow check, deferred: this rank's OWN header count (every rank checks its own: together that is the largest count of the step) travels to
    # pinned host memory -- one 4-byte copy, no kernel -- and is compared with this step's capacity at the START of the next exchange (a row past the
    # capacity is dropped by the pack kernel: with a capacity that belongs to this step it cannot happen -- if it ever does, gradients were lost and
    # the caller must hear about it)
    if dev.type == "cuda":
        host = getattr(arena, "_ovf_host", None)
        if host is None:
            host = arena._ovf_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        host.copy_(send[0, :1], non_blocking=True)
        oev = torch.cuda.Event()
        oev.record(torch.cuda.current_stream(dev))
        arena._ovf_pending = (oev, cap)
    else:                               # (CPU tensors, the gloo tests: nothing to wait for)
        arena._ovf_host = send[0, :1].clone()
        arena._ovf_pending = (None, cap)
    return {"allreduce": 4, "allgather": (1 + cap) * _C.GRAD_ROW_WORDS * 4, "rows": cap}


def _check_gather_overflow(arena) -> None:
    pend = getattr(arena, "_ovf_pending", None)
    if pend is None:
        return
    arena._ovf_pending = None
    oev, cap = pend
    if oev is not None:
        oev.synchronize()               # (recorded a whole step ago: returns at once)
    seen = int(arena._ovf_host[0])
    if seen > cap:
        _disarm_gather(arena)
        arena._gather_no_async = True   # (never armed again: this arena's capacities are agreed on synchronously from here on)
        raise RuntimeError(f"exchange_gradients(sparse='gather'): a rank touched {seen} rows in the previous step but the agreed capacity was {cap}: "
                           "gradient rows were dropped.  The ranks' backward / exchange sequences have diverged (every rank must run exactly one "
                           "backward into the arena per exchange); the capacity is agreed on synchronously from here on")


def overlap_factor_exchange(enable: bool = True) -> None:
    """Start the all-gather of the dL/dsh factors INSIDE the backward, as soon as the blend backward has produced them, so that it
    runs beside the per-Gaussian backward (the rasterizer's two-phase backward, options.backward_phase).  exchange_gradients then
    only waits for it.  Costs nothing with one rank."""
    from diff_gaussian_rasterization_ch3 import _C

    def hook(arena):
        if collectives_active() and getattr(arena, "world", 1) == dist.get_world_size():
            arena._gather_work = dist.all_gather_into_tensor(arena.gathered, arena.factor, async_op=True)
    _C.set_factor_ready_hook(hook if enable else None)


def reduce_densification_stats(point_grad_norm: torch.Tensor, visible_count: torch.Tensor,
                               max_radii: torch.Tensor) -> None:
    """In-place cross-rank reduction of the densification statistics of train.py:282-292:
    SUM of the screen-space gradient norms and visibility counts, MAX of the radii."""
    if collectives_active():
        dist.all_reduce(point_grad_norm, op=dist.ReduceOp.SUM)
        dist.all_reduce(visible_count, op=dist.ReduceOp.SUM)
        dist.all_reduce(max_radii, op=dist.ReduceOp.MAX)


def max_over_ranks(x: float, device: torch.device) -> float:
    if collectives_active():
        t = torch.tensor([x], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return float(x)


def barrier() -> None:
    if collectives_active():
        dist.barrier()


class StepBucket:
    """Gradient cache of one training step over an arbitrary set of leaves -- the reference's `_xyz_grd`, `_features_dc_grd`, ...,
    `mlp_grd[...]` (scene/saro_gaussian.py:226-265) as ONE flat fp32 buffer, so the cross-rank sum is one collective.  Built once
    for a fixed list of named leaves (the dynamic stage: six per-Gaussian groups + `_temporal_pos` = 60 floats per Gaussian, the
    four MLP heads and the hex-plane grids, saro_gaussian.py:306-319); rebuilt by the caller after densification changes P."""

    def __init__(self, leaves: Dict[str, torch.Tensor]):
        self.names = list(leaves.keys())
        self.leaves = [leaves[n] for n in self.names]
        dev = self.leaves[0].device if self.leaves else torch.device("cpu")
        self.flat = torch.zeros(sum(p.numel() for p in self.leaves), dtype=torch.float32, device=dev)
        self.views, o = [], 0
        for p in self.leaves:
            self.views.append(self.flat[o:o + p.numel()].view(p.shape))
            o += p.numel()

    def matches(self, leaves: Dict[str, torch.Tensor]) -> bool:
        """Is this bucket still the cache of `leaves` (the model's CURRENT parameters)?  Densification / pruning
        (densify_and_prune -> replace_tensor_to_optimizer / cat_tensors_to_optimizer, scene/saro_gaussian.py:560-640, :700-760) replaces
        the per-Gaussian parameters by NEW tensors with another P: the bucket then still points at the old ones, whose .grad no
        backward fills any more.  The training loop checks after every densification step and rebuilds:
            if not bucket.matches(model.leaves()): bucket = StepBucket(model.leaves())
        on every rank alike (densification is deterministic on replicated parameters)."""
        if list(leaves.keys()) != self.names:
            return False
        return all(q is p and tuple(q.shape) == tuple(v.shape) for q, p, v in zip(leaves.values(), self.leaves, self.views))

    def stale(self) -> bool:
        """A leaf was resized IN PLACE (p.data = ...) since the bucket was built: its cache view no longer fits."""
        return any(tuple(p.shape) != tuple(v.shape) for p, v in zip(self.leaves, self.views))

    def zero(self) -> None:                     # zero_gradient_cache (saro_gaussian.py:249-264)
        if self.stale():
            raise RuntimeError("StepBucket: a leaf changed shape since the bucket was built (densification?): rebuild it with "
                               "StepBucket(leaves) from the model's current parameters (see StepBucket.matches)")
        for p in self.leaves:                   # the previous step's .grad are views of this buffer: detach them first, or the
            p.grad = None                       # next backward would accumulate straight into the cache
        self.flat.zero_()

    def cache(self) -> None:
        """cache_gradient (saro_gaussian.py:226-247): add every leaf's .grad of the view just rendered; a leaf the view did not
        reach (grad None: the reference skips MLP weights without a gradient, :232) adds nothing.  Then optimizer.zero_grad(
        set_to_none=True) of train.py:221."""
        have = [(v, p.grad) for v, p in zip(self.views, self.leaves) if p.grad is not None]
        if have:
            torch._foreach_add_([v for v, _ in have], [g for _, g in have])
        for p in self.leaves:
            p.grad = None

    def partials(self, n: int) -> list:
        """Per-lane caches for distributed_step(views_in_flight=n): lane 0 is the cache itself, lanes 1.. are zeroed buffers of the
        same layout (kept for the next step)."""
        extra = getattr(self, "_extra", [])
        while len(extra) < n - 1:
            extra.append(torch.zeros_like(self.flat))
        self._extra = extra
        out = [self.views]
        for k in range(n - 1):
            extra[k].zero_()
            vs, o = [], 0
            for p in self.leaves:
                vs.append(extra[k][o:o + p.numel()].view(p.shape))
                o += p.numel()
            out.append(vs)
        return out

    def fold_partials(self, n: int) -> None:
        for k in range(n - 1):
            self.flat.add_(self._extra[k])

    def assign(self, ratio: float) -> None:     # set_batch_gradient (saro_gaussian.py:266-294): .grad = cache * (1 / batch)
        self.flat.mul_(ratio)
        for v, p in zip(self.views, self.leaves):
            p.grad = v


_LANE_STREAMS: Dict[tuple, list] = {}
_LANE_CONTEXTS: Dict[tuple, list] = {}


def _lane_contexts(dev: torch.device, n: int) -> list:
    """One rasterizer context per lane (diff_gaussian_rasterization_ch3._C.Context), made once per device: views in flight on different
    streams must not share a context's side streams, completion-pass gate and adaptive state (capacity hints, pose table)."""
    if dev.type != "cuda":
        return [contextlib.nullcontext()] * n
    from diff_gaussian_rasterization_ch3 import _C
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), n)
    if key not in _LANE_CONTEXTS:
        _LANE_CONTEXTS[key] = [_C.Context() for _ in range(n)]
    return _LANE_CONTEXTS[key]


def _lane_streams(dev: torch.device, n: int) -> list:
    """n side streams of a device, made once (a lane = one of the views in flight at a time)."""
    if dev.type != "cuda":
        return [None] * n
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), n)
    if key not in _LANE_STREAMS:
        _LANE_STREAMS[key] = [torch.cuda.Stream(device=dev) for _ in range(n)]
    return _LANE_STREAMS[key]


def distributed_step(bucket: StepBucket, views: Sequence, render_loss_fn, batch: Optional[int] = None,
                     views_in_flight: int = 1) -> Dict[str, torch.Tensor]:
    """One training iteration's gradient computation, sharded one view per rank -- the drop-in for the reference's sequential batch
    loop (train.py:190-226) with its gradient caching (scene/saro_gaussian.py:226-294) and densification statistics
    (train.py:279-292), same results up to fp32 summation order.

        for every view of this rank (views_of_rank):   out = render_loss_fn(view); out["loss"].backward()
            out: {"loss": scalar, "viewspace_points": the [P,3] means2D tensor whose .grad the rasterizer fills,
                  "visibility_filter": bool [P], "radii": int [P]}                       (getrenderparts, helper_train.py)
            statistics: ||viewspace_points.grad[:, :2]|| (train.py:212), visibility, radii                     -- per view
            bucket.cache()                                                               -- cache_gradient + zero_grad
        ONE all-reduce(SUM) of the flat gradient cache, started asynchronously, overlapped with the three small statistic
        reductions (SUM of gradient norms, SUM of visibility counts, MAX of radii); then leaf.grad = sum / batch
        (set_batch_gradient) -- the optimizer step that follows is the caller's, identical on every rank.

    Returns what train.py:279-292 derives for the densification: "visibility_count" [P], "visibility_filter" [P] (count > 0),
    "radii" [P] (max over the batch), "viewspace_point_grad" [P,1] (sum of the per-view norms / visibility count where
    visible), plus "loss" (mean over the batch, for logging).  `batch` defaults to len(views).

    views_in_flight = n > 1 (a rank that renders several views of the batch, i.e. batch > world size): consecutive views of this
    rank alternate between n streams, so that one view's binning (latency-bound) runs under the other's blend kernels (VALU-bound)
    -- measured on one MI355X at 1e6 Gaussians, 1080p: 1050 -> 1227 views/s with two views in flight (three: no better).  Each lane
    takes its gradients with torch.autograd.grad (nothing accumulates in the shared .grad fields) into its own partial cache; the
    partial caches and statistics are summed before the collectives.  Same results up to fp32 summation order.
    Memory: every view in flight holds its own rasterizer state until its backward has run -- ~320-350 B per Gaussian of geometry
    state (0.35 GB at 1e6 Gaussians, 0.95 GB at 3e6, gsrast_geometry_bytes), 4 B per listed instance + 20 B per column run of
    binning state (0.1-0.3 GB) and 19 MB per 1080p image -- plus one partial gradient cache (the size of the bucket) per extra lane."""
    multi = collectives_active()
    rank = dist.get_rank() if multi else 0
    world = dist.get_world_size() if multi else 1
    batch = len(views) if batch is None else batch
    bucket.zero()
    grad_norm = vis_count = max_radii = None
    loss_sum = None
    mine = list(views_of_rank(len(views), rank, world))
    if views_in_flight > 1 and len(mine) > 1:
        n = min(int(views_in_flight), len(mine))
        dev = bucket.flat.device
        streams = _lane_streams(dev, n)
        lane_ctx = _lane_contexts(dev, n)
        main = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
        partial = bucket.partials(n)                            # lane 0 adds into the cache itself
        stats = [None] * n
        for j, i in enumerate(mine):
            k = j % n
            ctx = torch.cuda.stream(streams[k]) if streams[k] is not None else contextlib.nullcontext()
            with ctx, lane_ctx[k]:
                if streams[k] is not None and j < n:
                    streams[k].wait_stream(main)                # the parameters (and the zeroed cache) are ordered on the caller's stream
                out = render_loss_fn(views[i])
                vsp = out["viewspace_points"]
                wanted = [p for p in bucket.leaves if p.requires_grad]
                grads = torch.autograd.grad(out["loss"], wanted + [vsp], allow_unused=True)
                pairs = [(v, g) for (v, p), g in zip([(v, p) for v, p in zip(partial[k], bucket.leaves) if p.requires_grad], grads[:-1]) if g is not None]
                if pairs:
                    torch._foreach_add_([v for v, _ in pairs], [g for _, g in pairs])
                gn = torch.norm(grads[-1][:, :2], dim=-1)                               # train.py:212
                vis = out["visibility_filter"].to(gn.dtype)
                rad = out["radii"].to(gn.dtype)
                ls = out["loss"].detach()
                st = stats[k]
                stats[k] = (gn, vis, rad, ls) if st is None else (st[0] + gn, st[1] + vis, torch.maximum(st[2], rad), st[3] + ls)
        for k in range(n):
            if streams[k] is not None:
                main.wait_stream(streams[k])
        bucket.fold_partials(n)
        for st in stats:
            if st is None:
                continue
            grad_norm = st[0] if grad_norm is None else grad_norm + st[0]
            vis_count = st[1] if vis_count is None else vis_count + st[1]
            max_radii = st[2] if max_radii is None else torch.maximum(max_radii, st[2])
            loss_sum = st[3] if loss_sum is None else loss_sum + st[3]
        mine = []
    for i in mine:
        out = render_loss_fn(views[i])
        out["loss"].backward()
        vsp = out["viewspace_points"]
        gn = torch.norm(vsp.grad[:, :2], dim=-1)                                    # train.py:212
        vis = out["visibility_filter"].to(gn.dtype)
        rad = out["radii"].to(gn.dtype)
        grad_norm = gn if grad_norm is None else grad_norm + gn
        vis_count = vis if vis_count is None else vis_count + vis
        max_radii = rad if max_radii is None else torch.maximum(max_radii, rad)
        loss_sum = out["loss"].detach() if loss_sum is None else loss_sum + out["loss"].detach()
        vsp.grad = None
        bucket.cache()
    if grad_norm is None:       # a rank without a view in this iteration (batch < world) still takes part in the collectives
        P = bucket.leaves[0].shape[0]
        z = torch.zeros(P, dtype=torch.float32, device=bucket.flat.device)
        grad_norm, vis_count, max_radii, loss_sum = z, z.clone(), z.clone(), torch.zeros((), device=bucket.flat.device)
    work = dist.all_reduce(bucket.flat, op=dist.ReduceOp.SUM, async_op=True) if multi else None
    reduce_densification_stats(grad_norm, vis_count, max_radii)
    if multi:
        dist.all_reduce(loss_sum, op=dist.ReduceOp.SUM)
        work.wait()
    bucket.assign(1.0 / batch)
    visible = vis_count > 0
    vgrad = grad_norm.clone()
    vgrad[visible] = vgrad[visible] / vis_count[visible]                            # train.py:286-287
    return {"visibility_count": vis_count, "visibility_filter": visible, "radii": max_radii,
            "viewspace_point_grad": vgrad.unsqueeze(1), "loss": loss_sum / batch}
