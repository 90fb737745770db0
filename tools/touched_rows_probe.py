#!/usr/bin/env python
"""How many Gaussians carry a non-zero gradient row per view, and in the union over the 8 views of a batch (what the sparse gradient
exchange moves, view_parallel.exchange_gradients(sparse=True)) -- one GPU, the views one after the other.
usage: tools/touched_rows_probe.py <P> <W> <H> [views]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "saro-gs_amd")]
import torch, bench, scenes
import diff_gaussian_rasterization_ch3 as rast
P, W, H = int(float(sys.argv[1])), int(sys.argv[2]), int(sys.argv[3])
V = int(sys.argv[4]) if len(sys.argv) > 4 else 8
dev = torch.device("cuda:0")
wl = bench.Workload(rast, scenes, P, W, H, 3, 0, V, dev, poses=V)
union = torch.zeros(P, dtype=torch.bool, device=dev)
per_view = []
for k in range(V):
    wl.step(None, 1)
    L = wl.leaves
    t = (L["means3D"].grad != 0).any(1) | (L["opacities"].grad != 0).any(1) | (L["scales"].grad != 0).any(1) | (L["rotations"].grad != 0).any(1) | (L["shs"].grad.reshape(P, -1) != 0).any(1)
    per_view.append(int(t.sum()))
    union |= t
n = int(union.sum())
dense = P * (44 * 2 * (V - 1) / V + 12 * (V - 1))
sparse = n * (44 * 2 * (V - 1) / V + 12 * (V - 1)) + P * 2 * (V - 1) / V
plain = P * 236 * 2 * (V - 1) / V
print(json.dumps(dict(P=P, W=W, H=H, views=V, touched_per_view=per_view, union=n, union_frac=round(n / P, 4),
                      link_MB_per_rank={"plain_allreduce_59_floats": round(plain / 1e6, 1), "factors_dense": round(dense / 1e6, 1), "factors_sparse": round(sparse / 1e6, 1)})))
