#!/usr/bin/env python
"""Development helper: the reference-shaped MLP heads (torch.nn, fp32) on 1 M rows, fwd+bwd, for rocprofv3."""
import sys, time
import torch, torch.nn as nn
dev = torch.device("cuda:0")
P, C, Hd, emb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, 32, 128, 9
mlp = lambda i, o: nn.Sequential(nn.Linear(i, Hd), nn.ReLU(), nn.Linear(Hd, Hd), nn.ReLU(), nn.Linear(Hd, o)).to(dev)
nets = [mlp(C + emb, 3), mlp(C + emb, 7), mlp(C + emb, 48)]
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "saro-gs_amd"))
if os.environ.get("SPLITK", "0") == "1":
    import fused_mlp
    nets = [fused_mlp.convert_heads(n) for n in nets]
x = torch.randn(P, C + emb, device=dev)
for it in range(6):
    if it == 1:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    for n in nets:
        for p in n.parameters(): p.grad = None
    sum(n(x).sum() for n in nets).backward()
torch.cuda.synchronize()
print("ms per fwd+bwd of the three 3-layer heads:", (time.perf_counter() - t0) / 5 * 1e3)
