#!/usr/bin/env python
"""Where the one-view-per-rank exchange's time goes on ONE GPU (a one-rank RCCL group, collectives forced): the rasterizer step with and without
the factor arena, and the pieces of view_parallel.exchange_gradients(sparse=True) bracketed with events (development helper)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "saro-gs_amd")]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
import numpy as np, torch, torch.distributed as dist
import bench, scenes, view_parallel as vp
import diff_gaussian_rasterization_ch3 as rast
_C = rast._C
P = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000
vp.force_collectives(True)
vp.init_from_env("nccl")
dev = torch.device("cuda:0")
wl = bench.Workload(rast, scenes, P, 1920, 1080, 3, 0, 8, dev, poses=8)

def timeit(fn, n=60):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

print("plain step (no arena)            %.3f ms" % timeit(lambda: wl.step(None, 1)))
arena = _C.GradArena(P, 16, dev, sh_factors=True, world=1)
_C.set_grad_arena(arena)
def fb():
    for p in list(wl.leaves.values()) + [wl.means2D]:
        p.grad = None
    arena.zero_grad()
    r = wl.rasters[wl.step_no % 8]; wl.step_no += 1
    color, _, _ = r(means3D=wl.leaves["means3D"], means2D=wl.means2D, opacities=wl.leaves["opacities"], shs=wl.leaves["shs"], scales=wl.leaves["scales"], rotations=wl.leaves["rotations"])
    color.backward(wl.g)
print("fwd + bwd into the factor arena  %.3f ms" % timeit(fb))
def fbx(sparse):
    fb()
    vp.exchange_gradients(arena, wl.leaves["means3D"].detach(), 1, sparse=sparse)
print("... + exchange dense factors     %.3f ms" % timeit(lambda: fbx(False)))
print("... + exchange sparse            %.3f ms" % timeit(lambda: fbx(True)))
print("... + exchange gather            %.3f ms" % timeit(lambda: fbx("gather")))
# pieces of the sparse exchange
fb(); torch.cuda.synchronize()
segs = arena.dense_segments(); fac = arena.factor[: 3 * P].view(P, 3)
touched = arena.touched.clone()
def piece(name, fn, n=30):
    print("   %-34s %.3f ms" % (name, timeit(fn, n)))
piece("all_reduce MAX of P bytes", lambda: dist.all_reduce(touched, op=dist.ReduceOp.MAX))
piece("nonzero (host sync)", lambda: torch.nonzero(touched, as_tuple=False).squeeze(1))
idx = torch.nonzero(touched, as_tuple=False).squeeze(1); n = idx.numel()
piece("cat(index_select x4)  n=%d" % n, lambda: torch.cat([sg.index_select(0, idx) for sg in segs], dim=1).contiguous())
comp = torch.cat([sg.index_select(0, idx) for sg in segs], dim=1).contiguous()
piece("all_reduce AVG [n, 11]", lambda: vp.allreduce_mean_inplace(comp.view(-1), 1))
def back():
    o = 0
    for sg in segs:
        sg.index_copy_(0, idx, comp[:, o: o + sg.shape[1]]); o += sg.shape[1]
piece("index_copy_ x4", back)
stride = ((3 * n + 3 + 3) // 4) * 4
def mine_():
    m = torch.zeros(stride, dtype=torch.float32, device=dev); m[: 3 * n] = fac.index_select(0, idx).reshape(-1); return m
piece("factor rows", mine_)
mine = mine_(); gathered = torch.empty(stride, dtype=torch.float32, device=dev)
piece("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(gathered, mine))
row_of = torch.full((P,), -1, dtype=torch.int32, device=dev); row_of[idx] = torch.arange(n, dtype=torch.int32, device=dev)
def rowof():
    row_of.fill_(-1); row_of[idx] = torch.arange(n, dtype=torch.int32, device=dev)
piece("row_of", rowof)
piece("sh_grad_combine (rows)", lambda: _C.sh_grad_combine(arena, wl.leaves["means3D"].detach(), gathered, 1, 1.0, rows=n, row_of=row_of, chunk_stride=stride))
piece("sh_grad_combine (union: clear + write)", lambda: _C.sh_grad_combine(arena, wl.leaves["means3D"].detach(), gathered, 1, 1.0, chunk_stride=stride, idx=idx))
# pieces of the all-gather exchange
fb(); torch.cuda.synchronize()
send = torch.empty((P + 1, 16), dtype=torch.int32, device=dev)
def hdr():
    send[0].zero_(); send[0, 1:4] = arena.factor[3 * P: 3 * P + 3].view(torch.int32)
piece("gather: header", hdr)
def pk():
    send[0].zero_(); _C.grad_rows_pack(arena, touched, send)
piece("gather: header zero + pack", pk)
cap = int(send[0, 0].item())
piece("gather: clone + all_reduce MAX (4 B)", lambda: dist.all_reduce(send[0, :1].clone(), op=dist.ReduceOp.MAX))
piece("gather: .item()", lambda: int(send[0, 0].item()))
mine = send[: 1 + cap]
g2 = torch.empty((1, 1 + cap, 16), dtype=torch.int32, device=dev)
piece("gather: all_gather_into_tensor cap=%d" % cap, lambda: dist.all_gather_into_tensor(g2.view(-1), mine.reshape(-1)))
piece("gather: clear sh rows (previous chunks)", lambda: _C.grad_rows_clear(arena, g2, dense=False, sh=True))
piece("gather: clear my dense rows", lambda: _C.grad_rows_clear(arena, mine.unsqueeze(0), dense=True, sh=False))
piece("gather: add one chunk", lambda: _C.grad_rows_add(arena, g2[0], wl.leaves["means3D"].detach(), 1.0))
dist.destroy_process_group()
