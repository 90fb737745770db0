"""CPU: the oracle's hand-derived backward (the formulas the HIP kernels also implement) against
central finite differences of the oracle's fp64 forward.

The rasterizer is only piecewise smooth (alpha < 1/255 and T < 1e-4 thresholds, integer radii, depth
order), so a probe that flips a discrete decision is discarded: the fp32 control state
(radii, point_list, n_contrib, the per-pixel signature of which list entries passed the alpha tests,
colour clamp flags) must be identical at x-eps and x+eps."""
import copy

import numpy as np
import pytest

PAIRS = {"means3D": "dL_dmeans3D", "scales": "dL_dscales", "rotations": "dL_drotations",
         "opacities": "dL_dopacity", "shs": "dL_dsh"}


@pytest.mark.parametrize("bg", [(0.0, 0.0, 0.0), (0.3, 0.6, 0.1)])
def test_backward_is_the_gradient_of_forward(orc, scenes, bg):
    P, W, H = 200, 48, 40
    sc = scenes.synth(P, 3, scale_mul=0.5)
    sc["bg"] = np.array(bg, np.float32)
    sc["opacities"] = (sc["opacities"] * 0.6).astype(np.float32)
    cam = scenes.camera(1, 5, W, H)
    g = scenes.upstream_grad(H, W, 4) * (H * W)
    base = orc.render(sc, cam, g, f64=True)

    def state(s):
        o = orc.render(s, cam, None, f64=True)
        return float((o["out_color"] * g).sum()), (o["radii"].tobytes(), o["point_list"].tobytes(), o["n_contrib"].tobytes(),
                                                        o["pair_hash"].tobytes(), o["clamped"].tobytes())

    _, ctl0 = state(sc)
    # The reference's backward is knowingly NOT the exact derivative for Gaussians whose centre lies
    # outside 1.3x the field of view: forward clamps t.x/t.z there (forward.cu:82-87) and the backward
    # only zeroes the x/y part (backward.cu:175-176, :262-263), ignoring d(clamp*t.z)/dt.z.  The oracle
    # restates that faithfully, so probe only unclamped Gaussians.
    V = cam["viewmatrix"].astype(np.float64)
    tv = sc["means3D"].astype(np.float64) @ V[:3, :3] + V[3, :3]
    unclamped = (np.abs(tv[:, 0] / tv[:, 2]) < 1.25 * cam["tanfovx"]) & (np.abs(tv[:, 1] / tv[:, 2]) < 1.25 * cam["tanfovy"])
    vis = np.nonzero((base["radii"] > 0) & unclamped)[0]
    rng = np.random.default_rng(0)
    for pname, gname in PAIRS.items():
        good = 0
        for _ in range(30):
            arr = sc[pname]
            idx = (int(rng.choice(vis)),) + tuple(int(rng.integers(0, s)) for s in arr.shape[1:])
            eps = np.float32(2.0 ** -12 * max(abs(float(arr[idx])), 0.05))
            s1, s2 = copy.deepcopy(sc), copy.deepcopy(sc)
            s1[pname][idx] = arr[idx] + eps
            s2[pname][idx] = arr[idx] - eps
            h = float(s1[pname][idx]) - float(s2[pname][idx])
            (l1, c1), (l2, c2) = state(s1), state(s2)
            if c1 != ctl0 or c2 != ctl0:
                continue                                  # a discrete decision flipped: not differentiable here
            fd = (l1 - l2) / h
            an = float(base[gname][idx])
            assert abs(fd - an) <= 2e-4 * max(1.0, abs(an)) + 1e-6, (pname, idx, fd, an)
            good += 1
        assert good >= 5, f"{pname}: too few smooth probes ({good})"
