"""TEST INFRASTRUCTURE ONLY: exact 3-nearest-neighbour mean squared distance in fp64 (scipy cKDTree), the quantity
`simple_knn._C.distCUDA2` returns (graphdeco-inria/simple-knn as shipped with 3DGS: `(best[0] + best[1] + best[2]) / 3`
over squared distances to the three nearest OTHER indices).  simple_knn is an un-vendored dependency of the reference
(/root/reference/scene/saro_gaussian.py:21) and is not in this container: parity unpinned against its binary; the search is
exact, so any exact 3-NN is the same function up to fp32 rounding."""
import numpy as np
from scipy.spatial import cKDTree


def mean_dist2(points: np.ndarray) -> np.ndarray:
    pts = np.asarray(points, np.float64)
    P = pts.shape[0]
    k = min(4, P)
    d, _ = cKDTree(pts).query(pts, k=k)          # column 0 is the point itself (distance 0)
    d2 = np.sort(d ** 2, axis=1)[:, 1:]
    return d2.sum(axis=1) / 3.0 if k == 4 else np.full(P, np.inf)
