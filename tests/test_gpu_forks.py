"""-m gpu: word forks.  In a file of its own, in front of test_gpu_gate.py: that file's two-thread soak switches the word forks off for the
rest of the process (their rule: one submitter at a time)."""
import sys
import os

import numpy as np  # noqa: F401
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu


def test_word_forks_change_nothing(scenes, rast, gpu):
    """The side stream forked by the next kernel's own start (a stored word + hipStreamWaitValue32) instead of an event: same
    colours, depths and radii bit for bit, same gradients up to the float atomics' order -- over poses the context knows and does not know,
    with the list cut (speculative launch, colour kernels behind the run emission) and the zero rows beside the blend backward in force."""
    import bench
    _C = rast._C
    P, W, H = 1_200_000, 1920, 1080
    wl = bench.Workload(rast, scenes, P, W, H, 3, view_k=0, n_views=6, dev=gpu, poses=3)
    for _ in range(6):
        wl.step(None, 1)
    if _C.get_option("concurrent_callers") or _C.get_option("stream_contexts") > 2:
        pytest.skip("an earlier test used the library from two threads at once: word forks are off for this process (their rule, DESIGN 4.4b)")
    outs = {}
    try:
        for wf in (1, 0, 1, 0):
            _C.set_option("word_fork", wf)
            res = []
            for _ in range(3):            # the three poses in turn
                raster = wl.rasters[wl.step_no % len(wl.rasters)]
                wl.step_no += 1
                for p in list(wl.leaves.values()) + [wl.means2D]:
                    p.grad = None
                L = wl.leaves
                color, radii, depth = raster(means3D=L["means3D"], means2D=wl.means2D, opacities=L["opacities"], shs=L["shs"], scales=L["scales"], rotations=L["rotations"])
                color.backward(wl.g)
                res.append((color.detach().clone(), depth.detach().clone(), radii.clone(), [L[k].grad.clone() for k in ("means3D", "opacities", "scales", "rotations", "shs")]))
            torch.cuda.synchronize()
            outs.setdefault(wf, []).append(res)
    finally:
        _C.set_option("word_fork", 0)          # (the library's default since round 6: opt-in)
    assert _C.context_query("last_late") > P // 8          # (the cut was in force)
    for a, b in zip(outs[1][0] + outs[1][1], outs[0][0] + outs[0][1]):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
        for ga, gb in zip(a[3], b[3]):
            assert bool(((ga - gb).abs() <= 1e-6 + 1e-3 * gb.abs()).all())
