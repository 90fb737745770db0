import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "saro-gs_amd"))
import torch, bench
dev = torch.device("cuda:0")
print(bench.knn_row(dev, 1_000_000))
