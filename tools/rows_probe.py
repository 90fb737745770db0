import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "saro-gs_amd"))
import torch, bench
import diff_gaussian_rasterization_ch3 as rast, scenes
dev = torch.device("cuda:0")
print(bench.iteration_row(rast, scenes, dev, 1_000_000, 1920, 1080, 3))
