"""CPU: the distCUDA2 oracle against an independent float64 k-d tree."""
import numpy as np
import pytest

from oracle import knn_oracle

scipy_spatial = pytest.importorskip("scipy.spatial")


def _kdtree_ref(pts):
    tree = scipy_spatial.cKDTree(pts.astype(np.float64))
    d, _ = tree.query(pts.astype(np.float64), k=4)      # self + 3 neighbours
    return (d[:, 1:] ** 2).mean(axis=1)


@pytest.mark.parametrize("kind", ["uniform", "clustered", "plane"])
def test_oracle_matches_kdtree(kind):
    rng = np.random.default_rng(3)
    P = 3000
    if kind == "uniform":
        pts = rng.uniform(-50, 50, (P, 3))
    elif kind == "clustered":
        c = rng.uniform(-100, 100, (20, 3))
        pts = c[rng.integers(0, 20, P)] + rng.normal(0, 0.5, (P, 3))
    else:
        pts = np.concatenate([rng.uniform(-30, 30, (P, 2)), np.full((P, 1), 1.5)], 1)
    pts = pts.astype(np.float32)
    got = knn_oracle.dist_cuda2(pts)
    ref = _kdtree_ref(pts)
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-9)


def test_oracle_duplicates_and_tiny_inputs():
    pts = np.array([[0, 0, 0], [0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 0]], np.float32)
    got = knn_oracle.dist_cuda2(pts)
    # point 0: two duplicates at distance 0 and (1,0,0) at 1 -> (0+0+1)/3
    assert got[0] == np.float32(1.0) / np.float32(3.0)
    assert got[2] == np.float32(1.0)                     # three points at squared distance 1
    # fewer than 4 points: FLT_MAX stays in the unused slots (simple_knn.cu:144,169-171)
    two = knn_oracle.dist_cuda2(np.array([[0, 0, 0], [3, 4, 0]], np.float32))
    assert np.isinf(two).all()
    assert knn_oracle.dist_cuda2(np.zeros((0, 3), np.float32)).shape == (0,)
