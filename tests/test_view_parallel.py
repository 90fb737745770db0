"""CPU, world_size 2 over gloo: the one exchange step of the multi-GPU path (flat-buffer gradient
all-reduce with batch-mean semantics, densification statistics) against the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "saro-gs_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    os.environ["OMP_NUM_THREADS"] = "1"          # eight ranks share this machine's cores (the oracle is OpenMP)
    import scenes
    import view_parallel as vp
    from oracle import oracle as orc
    r, l, w = vp.init_from_env("gloo")
    assert (r, w) == (rank, world)
    P, W, H = 400, 64, 48
    sc = scenes.synth(P, 3)
    my_views = vp.views_of_rank(world, rank, world)
    assert my_views == [rank]
    cam = scenes.camera(rank, world, W, H)            # one view per rank
    g = scenes.upstream_grad(H, W, 5) * (H * W)
    o = orc.render(sc, cam, g)                        # the oracle stands in for the GPU rasterizer on CPU
    names = ("dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dmeans2D")
    params = []
    for k in names:
        p = torch.zeros(o[k].shape, dtype=torch.float32)
        p.grad = torch.from_numpy(o[k].astype(np.float32)).clone()
        params.append(p)
    bucket = vp.FlatGradBucket(params)
    assert bucket.nbytes() == 4 * sum(int(np.prod(o[k].shape)) for k in names) == 4 * P * 62
    bucket.pack()
    bucket.allreduce_mean(batch=world)
    bucket.unpack()
    grad_norm = torch.from_numpy(np.linalg.norm(o["dL_dmeans2D"][:, :2], axis=1).astype(np.float32))
    vis = torch.from_numpy((o["radii"] > 0).astype(np.float32))
    radii = torch.from_numpy(o["radii"].astype(np.float32))
    vp.reduce_densification_stats(grad_norm, vis, radii)
    assert abs(vp.max_over_ranks(float(rank), torch.device("cpu")) - (world - 1)) < 1e-12
    # the zero-copy path of bench.py: one in-place mean over a flat buffer
    flat = torch.arange(8, dtype=torch.float32) * (rank + 1)
    vp.allreduce_mean_inplace(flat, world)
    assert torch.allclose(flat, torch.arange(8, dtype=torch.float32) * (sum(range(1, world + 1)) / world))
    vp.barrier()
    if rank == 0:
        np.savez(os.path.join(out_dir, "dist.npz"), grad_norm=grad_norm.numpy(), vis=vis.numpy(), radii=radii.numpy(),
                 **{k: p.grad.numpy() for k, p in zip(names, params)})
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8], ids=["two_ranks", "cfg4_eight_ranks_eight_views"])
def test_ranks_match_sequential_accumulation(tmp_path, orc, scenes, world):
    """world = 8: BASELINE.json configs[3]'s shape -- 8 views of one scene, one per rank, gradients averaged over the batch
    (scene/saro_gaussian.py:266-276) -- with the CPU oracle standing in for the rasterizer."""
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "dist.npz")
    # single process: the reference's loop -- render the views one after the other, sum, divide by batch
    P, W, H = 400, 64, 48
    sc = scenes.synth(P, 3)
    g = scenes.upstream_grad(H, W, 5) * (H * W)
    outs = [orc.render(sc, scenes.camera(k, world, W, H), g) for k in range(world)]
    for k in ("dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dmeans2D"):
        want = sum(o[k].astype(np.float32) for o in outs) / world
        np.testing.assert_allclose(got[k], want, rtol=1e-6, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(got["grad_norm"], sum(np.linalg.norm(o["dL_dmeans2D"][:, :2], axis=1) for o in outs), rtol=1e-5)
    np.testing.assert_array_equal(got["vis"], sum((o["radii"] > 0).astype(np.float32) for o in outs))
    np.testing.assert_array_equal(got["radii"], np.maximum.reduce([o["radii"] for o in outs]).astype(np.float32))


# ---- distributed_step: the reference's batch loop (train.py:190-226, saro_gaussian.py:226-294), one view per rank ----------------
class _DynamicStageModel(torch.nn.Module):
    """Leaves shaped like SaRO-GS's dynamic stage (saro_gaussian.py:306-319): six per-Gaussian groups + _temporal_pos (60 floats per
    Gaussian), MLP heads, hex-plane grids.  The "renderer" is a smooth torch stand-in (no rasterizer on CPU): what is under test is
    the step's bookkeeping, which never looks inside the render."""

    def __init__(self, P, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        r = lambda *s: torch.nn.Parameter(torch.randn(*s, generator=g) * 0.3)  # noqa: E731
        self._xyz, self._features_dc, self._features_rest = r(P, 3), r(P, 1, 3), r(P, 15, 3)
        self._opacity, self._scaling, self._rotation, self._temporal_pos = r(P, 1), r(P, 3), r(P, 4), r(P, 1)
        self.motion_mlp = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
        self.opacity_mlp = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.ReLU(), torch.nn.Linear(8, 1))
        self.unused_mlp = torch.nn.Linear(4, 4)            # never reached by a render: its .grad stays None (saro_gaussian.py:232)
        self.grids = torch.nn.ParameterList([r(4, 8, 8), r(4, 8, 8)])
        with torch.no_grad():
            for m in (self.motion_mlp, self.opacity_mlp, self.unused_mlp):
                for w in m.parameters():
                    w.copy_(torch.randn(w.shape, generator=g) * 0.2)

    def leaves(self):
        return dict(self.named_parameters())

    def render_loss(self, k):
        t = 0.37 * (k + 1)
        P = self._xyz.shape[0]
        plane = self.grids[k % 2]
        iu = (torch.arange(P) * 3 + k) % 8
        feat = plane[:, iu, (iu * 5 + 1) % 8].T                                     # [P, 4] "plane lookup"
        h = torch.cat([self._xyz, self._temporal_pos * t, feat], dim=1)            # [P, 8]
        pos = self._xyz + self.motion_mlp(h) * np.sin(t)
        means2D = torch.zeros((P, 3), requires_grad=True)
        screen = pos[:, :2] * np.cos(t) + means2D[:, :2]
        vis = (pos[:, 2].detach() + 0.1 * k) > -0.2
        radii = ((self._scaling.detach().exp().amax(1) * 7 + k).to(torch.int32)) * vis.to(torch.int32)
        w = torch.sigmoid(self._opacity[:, 0] + self.opacity_mlp(h)[:, 0]) * torch.exp(-(screen ** 2).sum(1)) * vis
        colour = self._features_dc[:, 0].sum(1) + self._features_rest.mean(1).sum(1) * np.cos(t)
        loss = (w * colour).sum() / P + 1e-2 * (self._rotation ** 2).sum() / P + 1e-2 * (self._scaling * t).sum() / P
        return {"loss": loss, "viewspace_points": means2D, "visibility_filter": vis, "radii": radii}


def _reference_loop(model, views):
    """train.py:190-226 + :279-292 spelled as the reference spells it."""
    leaves = model.leaves()
    cache = {n: torch.zeros_like(p) for n, p in leaves.items()}                    # zero_gradient_cache
    point_grad, visf, rads, loss_last = [], [], [], 0.0
    for k in views:
        out = model.render_loss(k)
        out["loss"].backward()
        point_grad.append(torch.norm(out["viewspace_points"].grad[:, :2], dim=-1))
        rads.append(out["radii"]); visf.append(out["visibility_filter"])
        for n, p in leaves.items():                                                # cache_gradient
            if p.grad is not None:
                cache[n] += p.grad.clone()
        for p in leaves.values():                                                  # optimizer.zero_grad(set_to_none=True)
            p.grad = None
        loss_last += float(out["loss"].detach())
    grads = {n: c * (1 / len(views)) for n, c in cache.items()}                    # set_batch_gradient
    count = torch.stack(visf, 1).sum(1)
    vfilter = count > 0
    radii = torch.stack(rads, 1).max(1)[0]
    g = torch.stack(point_grad, 1).sum(1)
    g[vfilter] = g[vfilter] / count[vfilter]
    return grads, dict(visibility_count=count, visibility_filter=vfilter, radii=radii, viewspace_point_grad=g.unsqueeze(1), loss=loss_last / len(views))


def _step_worker(rank, world, port, out_dir, n_views, in_flight=1):
    for p in (ROOT, os.path.join(ROOT, "saro-gs_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import view_parallel as vp
    torch.set_num_threads(1)
    vp.init_from_env("gloo")
    torch.manual_seed(0)
    model = _DynamicStageModel(257, 11)                    # replicated parameters: same seed on every rank
    bucket = vp.StepBucket(model.leaves())
    stats = vp.distributed_step(bucket, list(range(n_views)), model.render_loss, views_in_flight=in_flight)
    stats2 = vp.distributed_step(bucket, list(range(n_views)), model.render_loss, views_in_flight=in_flight)      # a second step starts from a clean cache
    for k in ("viewspace_point_grad", "radii", "visibility_count"):
        assert torch.equal(stats[k], stats2[k])
    if rank == 0:
        torch.save({"grads": {n: p.grad.clone() for n, p in model.leaves().items() if p.grad is not None},
                    "stats": {k: v.clone() for k, v in stats.items()}}, os.path.join(out_dir, "step.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_views", [(2, 2), (2, 4), (2, 3), (2, 1), (8, 8), (3, 8)],
                         ids=["one_view_per_rank", "two_views_per_rank", "uneven", "fewer_views_than_ranks",
                              "cfg4_eight_views_on_eight_ranks", "cfg4_eight_views_on_three_ranks"])
def test_distributed_step_equals_the_reference_batch_loop(tmp_path, world, n_views):
    """(8, 8) is BASELINE.json configs[3] (8 training views per iteration, one per GPU); (3, 8) the same batch on a node with fewer
    GPUs than views: ranks 0 and 1 render three views, rank 2 two."""
    mp.spawn(_step_worker, args=(world, _free_port(), str(tmp_path), n_views), nprocs=world, join=True)
    got = torch.load(tmp_path / "step.pt")
    model = _DynamicStageModel(257, 11)
    want_g, want_s = _reference_loop(model, list(range(n_views)))
    n_per_gaussian = sum(int(np.prod(p.shape[1:])) for n, p in model.leaves().items() if n.startswith("_"))
    assert n_per_gaussian == 60                                                    # the dynamic stage's per-Gaussian leaves
    assert set(got["grads"]) == set(want_g)                                        # every leaf gets a .grad, like set_batch_gradient
    for n in want_g:
        np.testing.assert_allclose(got["grads"][n].numpy(), want_g[n].numpy(), rtol=2e-5, atol=1e-7, err_msg=n)
    assert float(got["grads"]["unused_mlp.weight"].abs().max()) == 0.0
    assert float(got["grads"]["motion_mlp.0.weight"].abs().max()) > 0 and float(got["grads"]["grids.0"].abs().max()) > 0
    s = got["stats"]
    np.testing.assert_array_equal(s["visibility_count"].numpy(), want_s["visibility_count"].numpy().astype(np.float32))
    np.testing.assert_array_equal(s["visibility_filter"].numpy(), want_s["visibility_filter"].numpy())
    np.testing.assert_array_equal(s["radii"].numpy(), want_s["radii"].numpy().astype(np.float32))
    np.testing.assert_allclose(s["viewspace_point_grad"].numpy(), want_s["viewspace_point_grad"].numpy(), rtol=2e-5, atol=1e-9)
    assert abs(float(s["loss"]) - want_s["loss"]) < 1e-5


@pytest.mark.parametrize("n_views,in_flight", [(5, 2), (6, 3)], ids=["five_views_two_in_flight", "six_views_three_in_flight"])
def test_distributed_step_with_views_in_flight(tmp_path, n_views, in_flight):
    """Several views per rank, alternating between lanes (streams on a GPU; on the CPU the lanes only keep separate partial caches and
    take their gradients with autograd.grad): same gradients and statistics as the reference's sequential loop."""
    world = 2
    mp.spawn(_step_worker, args=(world, _free_port(), str(tmp_path), n_views, in_flight), nprocs=world, join=True)
    got = torch.load(tmp_path / "step.pt")
    model = _DynamicStageModel(257, 11)
    want_g, want_s = _reference_loop(model, list(range(n_views)))
    assert set(got["grads"]) == set(want_g)
    for n in want_g:
        np.testing.assert_allclose(got["grads"][n].numpy(), want_g[n].numpy(), rtol=2e-5, atol=1e-7, err_msg=n)
    s = got["stats"]
    np.testing.assert_array_equal(s["visibility_count"].numpy(), want_s["visibility_count"].numpy().astype(np.float32))
    np.testing.assert_array_equal(s["radii"].numpy(), want_s["radii"].numpy().astype(np.float32))
    np.testing.assert_allclose(s["viewspace_point_grad"].numpy(), want_s["viewspace_point_grad"].numpy(), rtol=2e-5, atol=1e-9)
    assert abs(float(s["loss"]) - want_s["loss"]) < 1e-5


# ---- overlapped factor exchange: the all-gather starts from the hook of the two-phase backward ---------------------------------
def _overlap_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "saro-gs_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import types
    import view_parallel as vp
    from diff_gaussian_rasterization_ch3 import _C
    vp.init_from_env("gloo")
    chunk = 16
    arena = types.SimpleNamespace(world=world, factor=torch.full((chunk,), float(rank + 1)), gathered=torch.zeros(world * chunk))
    assert _C._factor_ready_hook is None
    vp.overlap_factor_exchange(True)
    _C._factor_ready_hook(arena)                     # what rasterize_gaussians_backward does between the two phases
    work = arena._gather_work
    assert work is not None
    flat = torch.ones(4) * rank                      # a collective issued behind it (the dense all-reduce): same order on all ranks
    vp.allreduce_mean_inplace(flat, world)
    work.wait()
    want = torch.cat([torch.full((chunk,), float(r + 1)) for r in range(world)])
    ok = torch.equal(arena.gathered, want) and torch.allclose(flat, torch.ones(4) * (world - 1) / 2)
    # an arena built for another world size is left alone (exchange_gradients raises for it later)
    other = types.SimpleNamespace(world=world + 1, factor=arena.factor, gathered=arena.gathered)
    _C._factor_ready_hook(other)
    ok = ok and not hasattr(other, "_gather_work")
    vp.overlap_factor_exchange(False)
    ok = ok and _C._factor_ready_hook is None
    vp.barrier()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write(str(ok))
    dist.destroy_process_group()


def test_overlapped_factor_all_gather_two_ranks(tmp_path):
    world = 2
    mp.spawn(_overlap_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert [open(tmp_path / f"ok{r}").read() for r in range(world)] == ["True", "True"]


# ---- densification between two steps: P changes, the bucket is rebuilt -----------------------------------------------------------
def _densify(model, keep_every=3, clone_every=5):
    """A deterministic stand-in for densify_and_prune (scene/saro_gaussian.py:700-760): prune every `keep_every`-th Gaussian, clone
    every `clone_every`-th of the rest -- new Parameter objects with another P, as the reference's optimizer surgery leaves them."""
    P = model._xyz.shape[0]
    keep = torch.arange(P) % keep_every != 0
    idx = torch.nonzero(keep)[:, 0]
    idx = torch.cat([idx, idx[::clone_every]])
    for name in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_temporal_pos"):
        old = getattr(model, name)
        setattr(model, name, torch.nn.Parameter(old.detach()[idx].clone()))
    return int(idx.numel())


def _densify_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "saro-gs_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import view_parallel as vp
    torch.set_num_threads(1)
    vp.init_from_env("gloo")
    model = _DynamicStageModel(257, 11)
    bucket = vp.StepBucket(model.leaves())
    vp.distributed_step(bucket, [0, 1, 2, 3], model.render_loss)
    assert bucket.matches(model.leaves())
    P2 = _densify(model)                                    # the same surgery on every rank (replicated parameters)
    assert P2 != 257 and not bucket.matches(model.leaves())  # new Parameter objects: the old bucket caches tensors nobody trains any more
    # an in-place resize is caught by the step itself
    old_leaf = bucket.leaves[0]
    old_leaf.data = torch.zeros((5, 3))
    try:
        vp.distributed_step(bucket, [0, 1], model.render_loss)
        raised = False
    except RuntimeError as e:
        raised = "rebuild" in str(e)
    bucket = vp.StepBucket(model.leaves())
    stats = vp.distributed_step(bucket, [4, 5, 6, 7], model.render_loss)
    if rank == 0:
        torch.save({"raised": raised, "P2": P2, "grads": {n: p.grad.clone() for n, p in model.leaves().items() if p.grad is not None},
                    "stats": {k: v.clone() for k, v in stats.items()}}, os.path.join(out_dir, "densify.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_rebuilt_after_densification_changes_P(tmp_path):
    """train.py:279-300: every densification interval the per-Gaussian parameters are replaced by tensors with another P.  The step
    bucket detects it (refuses to run stale), is rebuilt from the new leaves, and the next distributed step equals the reference's
    batch loop on the densified model."""
    world = 2
    mp.spawn(_densify_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = torch.load(tmp_path / "densify.pt")
    assert got["raised"] is True
    model = _DynamicStageModel(257, 11)
    assert _densify(model) == got["P2"]
    want_g, want_s = _reference_loop(model, [4, 5, 6, 7])
    assert set(got["grads"]) == set(want_g)
    for n in want_g:
        assert got["grads"][n].shape == want_g[n].shape
        np.testing.assert_allclose(got["grads"][n].numpy(), want_g[n].numpy(), rtol=2e-5, atol=1e-7, err_msg=n)
    np.testing.assert_array_equal(got["stats"]["radii"].numpy(), want_s["radii"].numpy().astype(np.float32))
    np.testing.assert_allclose(got["stats"]["viewspace_point_grad"].numpy(), want_s["viewspace_point_grad"].numpy(), rtol=2e-5, atol=1e-9)


# ---- sparse factor exchange (round 4): only the rows some rank touched travel ----------------------------------------------------
def _sh_weights(dirs, deg):
    """w_k(direction) of forward.cu:20-71 / backward.cu:78-141 (what gsrast_sh_grad_combine_rows evaluates): [n, 16]."""
    x, y, z = dirs[:, 0], dirs[:, 1], dirs[:, 2]
    C0, C1 = 0.28209479177387814, 0.4886025119029199
    C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
    C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658, 1.445305721320277,
          -0.5900435899266435)
    w = torch.zeros((dirs.shape[0], 16), dtype=dirs.dtype)
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


def _sparse_worker(rank, world, port, out_dir, raw):
    for p in (ROOT, os.path.join(ROOT, "saro-gs_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    os.environ["OMP_NUM_THREADS"] = "1"
    import scenes
    import view_parallel as vp
    from diff_gaussian_rasterization_ch3 import _C
    from oracle import oracle as orc
    vp.init_from_env("gloo")
    P, W, H, deg, M = 600, 64, 48, 3, 16
    sc = scenes.synth(P, 13)
    cam = scenes.camera(rank, world, W, H)
    g = scenes.upstream_grad(H, W, 5) * (H * W)
    o = orc.render(sc, cam, g)                        # the oracle stands in for the GPU backward of this rank's view
    arena = _C.GradArena(P, M, torch.device("cpu"), sh_factors=True, world=world, raw=raw)
    names = ("xyz", "opacity_logit", "scaling", "rotation") if raw else ("means3D", "opacity", "scales", "rotations")
    for n, k in zip(names, ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations")):
        arena.take(n, (P, arena.widths[n]), False).copy_(torch.from_numpy(o[k].astype(np.float32)).reshape(P, -1))
    fac = torch.from_numpy((o["dL_dcolors"] * (1 - o["clamped"].astype(np.float32)) * (o["radii"] > 0)[:, None]).astype(np.float32))
    arena.factor[: 3 * P] = fac.reshape(-1)
    arena.factor[3 * P: 3 * P + 3] = torch.from_numpy(np.asarray(cam["campos"], np.float32))
    arena.last_degree = deg
    means = torch.from_numpy(sc["means3D"])
    calls = {}

    def combine(ar, means3D, chunks, n_views, scale, rows=None, row_of=None, chunk_stride=None, idx=None):      # the HIP kernel's arithmetic, in torch
        if idx is not None:             # (round 5: the union's rows as an index list; rows outside it are left as they are -- zero here)
            rows = int(idx.numel())
            row_of = torch.full((ar.P,), -1, dtype=torch.long)
            row_of[idx] = torch.arange(rows)
        rows = ar.P if rows is None else rows
        stride = ar.chunk if chunk_stride is None else chunk_stride
        row_of = torch.arange(ar.P) if row_of is None else row_of.long()
        acc = torch.zeros((ar.P, 16, 3))
        have = row_of >= 0
        for r in range(n_views):
            ch = chunks[r * stride: (r + 1) * stride]
            gr = ch[: 3 * rows].view(rows, 3)[row_of[have]]
            d = means3D[have] - ch[3 * rows: 3 * rows + 3]
            d = d / d.norm(dim=1, keepdim=True)
            acc[have] += _sh_weights(d, ar.last_degree)[:, :, None] * gr[:, None, :]
        acc *= scale
        calls.update(rows=rows, n_have=int(have.sum()))
        if getattr(ar, "raw", False):
            ar.take("features_dc", (ar.P, 1, 3), False).copy_(acc[:, :1])
            ar.take("features_rest", (ar.P, 15, 3), False).copy_(acc[:, 1:])
        else:
            ar.take("sh", (ar.P, 16, 3), False).copy_(acc)

    _C.sh_grad_combine = combine
    sent = vp.exchange_gradients(arena, means, world, sparse=True)
    # the literal mean of scene/saro_gaussian.py:266-276 over the batch: every rank's full gradients, summed / batch
    full = torch.cat([torch.from_numpy(o[k].astype(np.float32)).reshape(-1) for k in ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh")])
    dist.all_reduce(full)
    full /= world
    if raw:
        sh = torch.cat([arena.take("features_dc", (P, 1, 3), False), arena.take("features_rest", (P, 15, 3), False)], dim=1)
    else:
        sh = arena.take("sh", (P, 16, 3), False)
    got = torch.cat([arena.take(n, (P, arena.widths[n]), False).reshape(-1) for n in names] + [sh.reshape(-1)])
    err = float(((got - full).abs() / (1e-6 + 1e-4 * full.abs())).max())
    visible_somewhere = torch.from_numpy((o["radii"] > 0).astype(np.float32))
    dist.all_reduce(visible_somewhere, op=dist.ReduceOp.MAX)
    ok = err <= 1.0 and 0 < sent["rows"] <= int(visible_somewhere.sum()) and calls["rows"] == sent["rows"] == calls["n_have"] \
        and sent["allreduce"] == sent["rows"] * 44 + P and sent["allgather"] >= sent["rows"] * 12
    open(os.path.join(out_dir, f"ok{rank}"), "w").write(f"{ok} err {err:.3f} rows {sent['rows']} of {P}")
    vp.barrier()
    dist.destroy_process_group()


def _gather_worker(rank, world, port, out_dir, raw, uneven=False):
    """exchange_gradients(sparse="gather") with the three row kernels (csrc/gsrast_exchange.h) restated in torch.
    uneven: P = 4099 (no multiple of the pack kernel's 4096-Gaussian workgroups), rank 0 touches NO row, rank 1 touches EVERY row (so the
    step's capacity is P), the others what their view touches."""
    for p in (ROOT, os.path.join(ROOT, "saro-gs_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    os.environ["OMP_NUM_THREADS"] = "1"
    import scenes
    import view_parallel as vp
    from diff_gaussian_rasterization_ch3 import _C
    from oracle import oracle as orc
    vp.init_from_env("gloo")
    P, W, H, deg, M = (4099 if uneven else 600), 64, 48, 3, 16
    sc = scenes.synth(P, 13)
    arena = _C.GradArena(P, M, torch.device("cpu"), sh_factors=True, world=world, raw=raw)
    names = ("xyz", "opacity_logit", "scaling", "rotation") if raw else ("means3D", "opacity", "scales", "rotations")
    means = torch.from_numpy(sc["means3D"])
    counts = {"clear_dense": 0, "clear_sh": 0, "add": 0}

    def sh_views(ar):
        if getattr(ar, "raw", False):
            return [ar.take("features_dc", (ar.P, 3), False), ar.take("features_rest", (ar.P, 45), False)]
        return [ar.take("sh", (ar.P, 48), False)]

    def rows_of(chunk):
        n = int(chunk[0, 0])
        return chunk[1: 1 + n]

    def pack(ar, touched, rows):
        idx = torch.nonzero(touched).squeeze(1)
        n = idx.numel()
        body = torch.cat([idx.to(torch.int32).view(-1, 1), torch.cat(ar.dense_segments(), 1)[idx].contiguous().view(torch.int32),
                          ar.factor[: 3 * ar.P].view(ar.P, 3)[idx].contiguous().view(torch.int32), torch.zeros((n, 1), dtype=torch.int32)], 1)
        rows[1: 1 + n] = body
        rows[0, 0] += n

    def clear(ar, chunks, dense, sh):
        counts["clear_dense" if dense else "clear_sh"] += 1
        for ch in chunks:
            idx = rows_of(ch)[:, 0].long()
            for v in (ar.dense_segments() if dense else []) + (sh_views(ar) if sh else []):
                v[idx] = 0.0

    def add(ar, chunk, means3D, scale):
        counts["add"] += 1
        r = rows_of(chunk)
        idx = r[:, 0].long()
        o = 1
        for sg in ar.dense_segments():
            sg[idx] += scale * r[:, o: o + sg.shape[1]].contiguous().view(torch.float32)
            o += sg.shape[1]
        g = r[:, 12:15].contiguous().view(torch.float32)
        d = means3D[idx] - chunk[0, 1:4].contiguous().view(torch.float32)
        d = d / d.norm(dim=1, keepdim=True)
        term = (_sh_weights(d, ar.last_degree)[:, :, None] * g[:, None, :] * scale).reshape(-1, 48)
        vs = sh_views(ar)
        if len(vs) == 1:
            vs[0][idx] += term
        else:
            vs[0][idx] += term[:, :3]
            vs[1][idx] += term[:, 3:]

    _C.grad_rows_pack, _C.grad_rows_clear, _C.grad_rows_add = pack, clear, add
    ok, report = True, []
    for step in range(3):               # three steps with different views: a row that leaves the union must read zero again
        cam = scenes.camera((rank + 3 * step) % 11, 11, W, H)
        o = orc.render(sc, cam, scenes.upstream_grad(H, W, 5 + step) * (H * W))
        for n, k in zip(names, ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations")):
            arena.take(n, (P, arena.widths[n]), False).copy_(torch.from_numpy(o[k].astype(np.float32)).reshape(P, -1))
        fac = torch.from_numpy((o["dL_dcolors"] * (1 - o["clamped"].astype(np.float32)) * (o["radii"] > 0)[:, None]).astype(np.float32))
        if uneven and rank == 0:        # a view that touched nothing: all-zero rows, count 0
            for n in names:
                arena.take(n, (P, arena.widths[n]), False).zero_()
            fac = torch.zeros_like(fac)
            o = dict(o, **{k: np.zeros_like(o[k]) for k in ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh")})
        if uneven and rank == 1:        # a view that touched EVERY row: count = P = the step's capacity
            gen = torch.Generator().manual_seed(100 + step)
            o = dict(o)
            for n, k in zip(names, ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations")):
                v = torch.randn((P, arena.widths[n]), generator=gen) + 3.0       # (no zero anywhere)
                arena.take(n, (P, arena.widths[n]), False).copy_(v)
                o[k] = v.numpy()
            fac = torch.randn((P, 3), generator=gen) + 3.0
            d = means - torch.from_numpy(np.asarray(cam["campos"], np.float32))
            d = d / d.norm(dim=1, keepdim=True)
            o["dL_dsh"] = (_sh_weights(d, deg)[:, :, None] * fac[:, None, :]).numpy()
        arena.factor[: 3 * P] = fac.reshape(-1)
        arena.factor[3 * P: 3 * P + 3] = torch.from_numpy(np.asarray(cam["campos"], np.float32))
        arena.last_degree = deg
        sent = vp.exchange_gradients(arena, means, world, sparse="gather")
        if uneven:
            ok = ok and sent["rows"] == P
        full = torch.cat([torch.from_numpy(o[k].astype(np.float32)).reshape(-1) for k in ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh")])
        dist.all_reduce(full)
        full /= world
        if raw:
            sh = torch.cat([arena.take("features_dc", (P, 1, 3), False), arena.take("features_rest", (P, 15, 3), False)], dim=1)
        else:
            sh = arena.take("sh", (P, 16, 3), False)
        got = torch.cat([arena.take(n, (P, arena.widths[n]), False).reshape(-1) for n in names] + [sh.reshape(-1)])
        err = float(((got - full).abs() / (1e-6 + 1e-4 * full.abs())).max())
        same = got.clone()
        dist.broadcast(same, src=0)
        mine = int((torch.from_numpy(o["radii"]) > 0).sum())
        ok = ok and err <= 1.0 and torch.equal(same, got) and 0 < sent["rows"] <= P and sent["allgather"] == (1 + sent["rows"]) * 64 and sent["allreduce"] == 4 \
            and sent["rows"] >= 1 and (mine >= 1 or uneven)
        report.append(f"step {step} err {err:.3f} rows {sent['rows']} same {torch.equal(same, got)}")
    ok = ok and counts["add"] == 3 * world and counts["clear_dense"] == 3 and counts["clear_sh"] == 2      # (the first step zeroes the SH region whole)
    open(os.path.join(out_dir, f"ok{rank}"), "w").write(f"{ok} {report} {counts}")
    vp.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("raw", [False, True], ids=["rasterizer_leaves", "raw_leaves"])
@pytest.mark.parametrize("world", [2, 8], ids=["two_ranks", "cfg4_eight_ranks"])
def test_all_gather_exchange_equals_the_literal_batch_mean_bit_identical_on_every_rank(tmp_path, world, raw):
    """exchange_gradients(sparse="gather") on CPU over gloo: ONE all-gather of the 64-byte rows each rank's own view touched (cap from a
    4-byte MAX all-reduce), the chunks added in rank order -- the reference's batch mean (scene/saro_gaussian.py:266-276) within
    tolerance, the SAME BITS on every rank, over three steps whose views differ (rows that leave the union read zero again)."""
    mp.spawn(_gather_worker, args=(world, _free_port(), str(tmp_path), raw), nprocs=world, join=True)
    reports = [open(tmp_path / f"ok{r}").read() for r in range(world)]
    assert all(r.startswith("True") for r in reports), reports


def test_all_gather_exchange_with_strongly_uneven_counts_eight_ranks(tmp_path):
    """The same exchange over 8 gloo ranks with P = 4099 (not a multiple of the pack kernel's 4096-Gaussian workgroups), one rank whose view
    touched NO row (count 0: an empty chunk behind its header) and one that touched EVERY row (count = P = the step's capacity):
    the literal batch mean (scene/saro_gaussian.py:266-276), the same bits on every rank, three steps."""
    mp.spawn(_gather_worker, args=(8, _free_port(), str(tmp_path), False, True), nprocs=8, join=True)
    reports = [open(tmp_path / f"ok{r}").read() for r in range(8)]
    assert all(r.startswith("True") for r in reports), reports


class _FakeArena:
    pass


def test_gather_capacity_event_is_only_taken_from_the_last_backward_and_overflow_is_loud():
    """(a) a capacity agreed on beside ANOTHER backward than the step's last one is not used (sequence numbers); (b) any other
    exchange form disarms the hook for good; (c) a header count above the agreed capacity -- gradient rows dropped -- raises at the start of
    the next exchange instead of passing silently."""
    import view_parallel as vp

    class Ev:
        def __init__(self): self.waited = 0
        def synchronize(self): self.waited += 1

    a = _FakeArena()
    a._gather_armed, a._cap_event, a.touched_reader_event = True, Ev(), object()
    ev = a._cap_event
    vp._disarm_gather(a)
    assert a._gather_armed is False and a._cap_event is None and a.touched_reader_event is None and ev.waited == 1
    # (c) the deferred overflow check
    b = _FakeArena()
    b._ovf_host, b._ovf_pending, b._gather_armed = torch.tensor([17], dtype=torch.int32), (None, 16), True
    with pytest.raises(RuntimeError, match="rows were dropped"):
        vp._check_gather_overflow(b)
    assert b._gather_armed is False and b._ovf_pending is None and b._gather_no_async is True
    b._ovf_host, b._ovf_pending = torch.tensor([16], dtype=torch.int32), (None, 16)
    vp._check_gather_overflow(b)        # count == capacity: fine
    # (a) is a property of _exchange_gather's own code path: the event's tag against the arena's backward counter
    import inspect
    src = inspect.getsource(vp._exchange_gather)
    assert "_cap_seq" in src and "touched_seq" in src


@pytest.mark.parametrize("raw", [False, True], ids=["rasterizer_leaves", "raw_leaves"])
@pytest.mark.parametrize("world", [2, 8], ids=["two_ranks", "cfg4_eight_ranks"])
def test_sparse_factor_exchange_equals_the_literal_batch_mean(tmp_path, world, raw):
    """exchange_gradients(sparse=True) on CPU over gloo, the oracle as each rank's backward and the combine kernel's arithmetic
    restated in torch: the union of touched rows is agreed on with a MAX all-reduce of one byte per Gaussian, only those rows are
    all-reduced (11 floats) / all-gathered (3 floats), and every leaf gradient -- the rasterizer's own five, or the six RAW leaves
    of GaussianRasterizerRaw with dL/dsh split into features_dc / features_rest -- equals the reference's batch mean
    (scene/saro_gaussian.py:266-276); the bytes handed to the collectives follow the row count."""
    mp.spawn(_sparse_worker, args=(world, _free_port(), str(tmp_path), raw), nprocs=world, join=True)
    reports = [open(tmp_path / f"ok{r}").read() for r in range(world)]
    assert all(r.startswith("True") for r in reports), reports
