#!/usr/bin/env python
"""Development helper: the hexplane bench row alone (python tools/hexplane_probe.py [P] [scatter 0|1])."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "saro-gs_amd"))
import torch
import bench
from diff_gaussian_rasterization_ch3 import _C
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
if len(sys.argv) > 2:
    _C.lib().gsrast_set_option(b"hexplane_scatter", int(sys.argv[2]))
print(json.dumps(bench.hexplane_row(torch.device("cuda:0"), P), indent=1))
