import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "saro-gs_amd"))
import torch, bench
dev = torch.device("cuda:0")
for P in (250_000, 1_000_000, 4_000_000):
    print(P, bench.epilogue_row(dev, P))
