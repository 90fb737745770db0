"""CPU: unit checks of the oracle's building blocks and of the invariants the binning must obey."""
import numpy as np
import pytest


def test_fixed_sequence_exp_is_a_faithful_expf(orc):
    """exp_mode 0 is a legitimate expf: <= 2 ulp from the correctly rounded value over the whole
    range the blend kernels can feed it (power in [-80, 0]; alpha is thresholded at 1/255)."""
    orc.set_exp_mode(0)
    rng = np.random.default_rng(0)
    xs = np.concatenate([np.linspace(-12, 0, 4001), rng.uniform(-80, 0, 4000), [-80.0, -79.99, 0.0, -1e-8, -5.541264]]).astype(np.float32)
    got = np.array([orc.expf(float(x)) for x in xs], dtype=np.float32)
    want = np.exp(xs.astype(np.float64))
    ulp = np.spacing(want.astype(np.float32)).astype(np.float64)
    assert (np.abs(got.astype(np.float64) - want) <= 2.0 * ulp + 1e-45).all()
    assert orc.expf(-81.0) == 0.0 and orc.expf(0.0) == 1.0
    orc.set_exp_mode(1)
    try:
        got1 = np.array([orc.expf(float(x)) for x in xs[:200]], dtype=np.float32)
        assert (np.abs(got1.astype(np.float64) - want[:200]) <= 1.0 * ulp[:200] + 1e-45).all()
    finally:
        orc.set_exp_mode(0)


@pytest.mark.parametrize("n,want", [(1, 1), (2, 2), (3, 2), (625, 10), (2500, 12), (5440, 13), (8160, 13), (65536, 17)])
def test_higher_msb(orc, n, want):
    """getHigherMsb (rasterizer_impl.cu:35-50): SURVEY.md 8 lists 10 / 12 / 13 / 13 key bits for cfg1/2/3/5."""
    assert orc.higher_msb(n) == want


@pytest.mark.parametrize("P,W,H", [(500, 64, 48), (3000, 200, 136), (1, 16, 16)])
def test_binning_invariants(orc, scenes, P, W, H):
    sc = scenes.synth(P, 5)
    cam = scenes.camera(1, 3, W, H)
    o = orc.forward(sc, cam)
    R = o["R"]
    assert R == int(o["tiles_touched"].astype(np.int64).sum()) == int(o["point_offsets"][-1])
    keys = o["keys_sorted"]
    assert (np.diff(keys.astype(np.uint64)) >= 0).all() if R > 1 else True          # sortedness
    assert sorted(o["keys_unsorted"].tolist()) == keys.tolist()                     # permutation
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    T = o["ranges"].shape[0]
    cover = np.zeros(R, np.int32)
    for t in range(T):
        a, b = o["ranges"][t]
        assert a <= b
        cover[a:b] += 1
        assert (tiles[a:b] == t).all()
        # inside a tile: depth ascending, ties broken by Gaussian index (stable sort)
        d = o["depths"][o["point_list"][a:b]]
        assert (np.diff(d) >= 0).all()
        same = np.diff(d) == 0
        assert (np.diff(o["point_list"][a:b].astype(np.int64))[same] > 0).all()
    assert (cover == 1).all()                                                       # ranges partition [0, R)
    # key low bits are the depth's float bits
    np.testing.assert_array_equal((keys & np.uint64(0xFFFFFFFF)).astype(np.uint32), o["depths"][o["point_list"]].view(np.uint32))
    assert (o["n_contrib"] <= (o["ranges"][:, 1] - o["ranges"][:, 0]).max()).all()


def test_fp64_truth_build_agrees_with_fp32(orc, scenes):
    sc = scenes.synth(4000, 9)
    cam = scenes.camera(0, 1, 240, 160)
    g = scenes.upstream_grad(160, 240, 10)
    a, b = orc.render(sc, cam, g), orc.render(sc, cam, g, f64=True)
    np.testing.assert_array_equal(a["n_contrib"], b["n_contrib"])
    np.testing.assert_array_equal(a["radii"], b["radii"])
    for k in ("out_color", "out_depth", "final_T"):
        assert np.abs(a[k] - b[k]).max() < 1e-5
    for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dcov3D"):
        assert np.abs(a[k] - b[k]).max() < 1e-5, k


def test_empty_and_culled_inputs(orc, scenes):
    cam = scenes.camera(0, 1, 48, 32)
    sc = scenes.synth(16, 1)
    sc["means3D"][:] = np.array([50.0, 0.0, 0.0], np.float32)     # all outside the view
    sc["bg"] = np.array([0.2, 0.4, 0.6], np.float32)
    o = orc.render(sc, cam, np.ones((3, 32, 48), np.float32))
    assert o["R"] == 0 and not o["radii"].any()
    for c in range(3):
        assert (o["out_color"][c] == sc["bg"][c]).all()
    assert (o["out_depth"] == 15.0).all() and not o["dL_dmeans3D"].any()
