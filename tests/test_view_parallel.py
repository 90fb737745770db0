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


def test_two_ranks_match_sequential_accumulation(tmp_path, orc, scenes):
    world = 2
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
    np.testing.assert_array_equal(got["radii"], np.maximum(outs[0]["radii"], outs[1]["radii"]).astype(np.float32))
