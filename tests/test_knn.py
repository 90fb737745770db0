"""simple_knn._C.distCUDA2 (second item).  CPU: the oracle on hand-checkable inputs.
GPU (-m gpu): the HIP implementation against the exact fp64 3-NN oracle."""
import numpy as np
import pytest
import torch


def test_oracle_known_answers():
    from oracle import knn_oracle
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3], [10, 10, 10]], np.float64)
    got = knn_oracle.mean_dist2(pts)
    np.testing.assert_allclose(got[0], (1 + 4 + 9) / 3.0)
    np.testing.assert_allclose(got[1], (1 + 5 + 10) / 3.0)
    np.testing.assert_allclose(got[4], (300 - 60 + 9 + 300 - 40 + 4 + 300 - 20 + 1) / 3.0)   # to (0,0,3), (0,2,0), (1,0,0)


def _clouds():
    rng = np.random.default_rng(7)
    yield "uniform_5k", rng.uniform(-1.3, 1.3, size=(5000, 3))
    yield "clustered", np.concatenate([rng.normal(c, 0.01, size=(700, 3)) for c in rng.uniform(-5, 5, size=(9, 3))])
    dup = rng.uniform(0, 1, size=(1500, 3)); dup[500:1000] = dup[:500]                       # exact duplicates: distance 0 counts
    yield "duplicates", dup
    line = np.zeros((3000, 3)); line[:, 0] = np.sort(rng.uniform(0, 100, 3000))              # degenerate bounding box (two flat axes)
    yield "collinear", line
    yield "tiny_4", rng.normal(size=(4, 3))
    yield "uniform_200k", rng.uniform(-50, 50, size=(200_000, 3))
    yield "boxes_edge_1025", rng.uniform(0, 1, size=(1025, 3))


@pytest.mark.gpu
@pytest.mark.parametrize("name,pts", list(_clouds()), ids=[n for n, _ in _clouds()])
def test_hip_knn_matches_exact_oracle(name, pts, gpu):
    from oracle import knn_oracle
    from simple_knn._C import distCUDA2
    p32 = pts.astype(np.float32)
    got = distCUDA2(torch.from_numpy(p32).to(gpu)).cpu().numpy().astype(np.float64)
    want = knn_oracle.mean_dist2(p32.astype(np.float64))
    # fp32 squared distances: relative 1e-6 of the coordinates' scale squared
    scale2 = float(np.abs(p32).max()) ** 2
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=4e-7 * scale2)
    # and it feeds the reference's initialisation as written (scene/saro_gaussian.py:187-189)
    d2 = torch.clamp_min(torch.from_numpy(got), 0.0000001)
    assert torch.isfinite(torch.log(torch.sqrt(d2))).all()


@pytest.mark.gpu
def test_hip_knn_argument_errors(gpu):
    from simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(10, 3))                      # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(10, 2, device=gpu))
    assert distCUDA2(torch.zeros(0, 3, device=gpu)).numel() == 0
