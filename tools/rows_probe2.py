#!/usr/bin/env python
"""Development helper: the dynamic-stage iteration row alone."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "saro-gs_amd"))
import torch
import bench
import diff_gaussian_rasterization_ch3 as rast
import scenes
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
print(json.dumps(bench.dynamic_iteration_row(rast, scenes, torch.device("cuda:0"), P, 1920, 1080, 3), indent=1))
