#!/usr/bin/env python
"""Development helper: a few fwd+bwd passes of the fused hexplane field only (for rocprofv3 --kernel-trace --stats).
python tools/hexplane_prof.py [dnerf|neural3d] [P]"""
import itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "saro-gs_amd"))
import numpy as np
import torch
import fused_hexplane
which = sys.argv[1] if len(sys.argv) > 1 else "dnerf"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
reso = [64, 64, 64, 128] if which == "dnerf" else [512, 512, 512, 256]
dev = torch.device("cuda:0")
from diff_gaussian_rasterization_ch3 import _C
_C.lib().gsrast_set_option(b"ablate", int(os.environ.get("HEX_ABLATE", "0")))
g = torch.Generator(device="cpu").manual_seed(0)
coo = list(itertools.combinations(range(4), 2))
grids = [torch.randn((1, 32, reso[b], reso[a]), generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True) for (a, b) in coo]
pts = torch.rand((P, 4), generator=g).to(dev)
levels = torch.cat([torch.rand((P, 3), generator=g) * float(np.log2(reso[0])), torch.zeros((P, 1))], dim=1).to(dev)
dy = torch.randn((P, 32), generator=g).to(dev)
for _ in range(6):
    for gr in grids: gr.grad = None
    o = fused_hexplane.interpolate_ms_features(pts, [grids], 2, True, levels, None)
    o.backward(dy)
torch.cuda.synchronize()
print("done")
