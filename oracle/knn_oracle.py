"""CPU restatement of simple_knn's distCUDA2 -- TEST INFRASTRUCTURE ONLY.

Reference: submodules/simple-knn/simple_knn.cu:136-182 (boxMeanDist) and spatial.cu:14-25: for every
point the mean of the squared distances to its 3 nearest OTHER points.  The reference's Morton
boxes only prune (a box is skipped when its distance exceeds the running 3rd-best, simple_knn.cu:
163-167), so the result is the exact 3-NN multiset whatever the traversal order; this oracle is
therefore plain brute force.  Arithmetic restated in fp32, one rounding per operation, in the
reference's order:  d = (p - q);  dist = d.x*d.x + d.y*d.y + d.z*d.z  (simple_knn.cu:123-124);
result = (best0 + best1 + best2) / 3.0f  with best0 <= best1 <= best2 (simple_knn.cu:181).
Fewer than 4 points leave FLT_MAX in the unused slots, exactly like the reference.

Parity unpinned: the reference is CUDA-only (cub/thrust), nothing of it can be executed here.  The
oracle is checked against an independent float64 k-d tree instead (tests/test_knn_oracle.py).
"""
import numpy as np

FLT_MAX = np.float32(3.4028234663852886e38)


def dist_cuda2(points: np.ndarray, chunk: int = 512) -> np.ndarray:
    pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
    P = pts.shape[0]
    out = np.empty(P, np.float32)
    with np.errstate(over="ignore"):
        for s in range(0, P, chunk):
            q = pts[s:s + chunk]                                  # [c,3]
            d = pts[None, :, :] - q[:, None, :]                   # point - ref, fp32
            dist = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]  # fp32, left to right
            dist = dist + d[..., 2] * d[..., 2]
            dist[np.arange(q.shape[0]), np.arange(s, s + q.shape[0])] = np.inf   # i == idx skipped
            k = min(3, P - 1)
            best = np.full((q.shape[0], 3), FLT_MAX, np.float32)
            if k > 0:
                part = np.partition(dist, k - 1, axis=1)[:, :k]
                part.sort(axis=1)
                best[:, :k] = part
            out[s:s + chunk] = (best[:, 0] + best[:, 1] + best[:, 2]) / np.float32(3.0)
    return out
