"""GPU parity of distCUDA2 (gaussianrpg_amd/csrc/knn.hip) against oracle/knn_oracle.py."""
import numpy as np
import pytest
import torch

from oracle import knn_oracle

pytestmark = pytest.mark.gpu


def _run(pts_np):
    from simple_knn._C import distCUDA2
    out = distCUDA2(torch.from_numpy(pts_np).cuda())
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _cloud(kind, P, seed):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        p = rng.uniform(-40, 40, (P, 3))
    elif kind == "clustered":
        c = rng.uniform(-100, 100, (32, 3))
        p = c[rng.integers(0, 32, P)] + rng.normal(0, 0.3, (P, 3))
    elif kind == "street":      # ground plane + facades, like the lidar initialisation
        p = np.concatenate([rng.uniform(-30, 30, (P, 1)), rng.normal(1.5, 0.02, (P, 1)),
                            rng.uniform(-20, 300, (P, 1))], 1)
    else:                        # duplicates: every point appears about four times
        base = rng.uniform(-5, 5, (max(P // 4, 1), 3))
        p = base[rng.integers(0, base.shape[0], P)]
    return p.astype(np.float32)


@pytest.mark.parametrize("kind", ["uniform", "clustered", "street", "dups"])
@pytest.mark.parametrize("P", [5, 257, 1024, 1025, 6000])
def test_bit_exact_vs_oracle(kind, P):
    pts = _cloud(kind, P, P)
    got = _run(pts)
    ref = knn_oracle.dist_cuda2(pts)
    np.testing.assert_array_equal(got, ref)     # exact 3-NN, same fp32 arithmetic: bit-identical


def test_tiny_inputs_like_reference():
    # fewer than 4 points: FLT_MAX stays in the unused slots (simple_knn.cu:144): +inf for 1-2
    # points (the sum overflows), FLT_MAX/3 for 3 points
    for P in (1, 2, 3):
        pts = _cloud("uniform", P, 7)
        got = _run(pts)
        np.testing.assert_array_equal(got, knn_oracle.dist_cuda2(pts))
        assert (got > 1e37).all()
    got = _run(np.zeros((0, 3), np.float32))
    assert got.shape == (0,)
    four = _cloud("uniform", 4, 8)
    np.testing.assert_array_equal(_run(four), knn_oracle.dist_cuda2(four))


def test_large_cloud_against_kdtree():
    scipy_spatial = pytest.importorskip("scipy.spatial")
    pts = _cloud("street", 300_000, 11)
    got = _run(pts)
    d, _ = scipy_spatial.cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    ref = (d[:, 1:] ** 2).mean(axis=1)
    np.testing.assert_allclose(got, ref, rtol=5e-5, atol=1e-9)


def test_rejects_cpu_tensor():
    from simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(8, 3))
