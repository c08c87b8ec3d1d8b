"""``simple_knn._C``: the one op the reference binds (submodules/simple-knn/ext.cpp:15-17)."""
from gaussianrpg_amd import _C as _grpg_C


def distCUDA2(points):
    """[P,3] float device tensor -> [P] mean squared distance to the 3 nearest other points
    (submodules/simple-knn/spatial.cu:14-25).  No CPU path: CPU tensors are rejected."""
    return _grpg_C.distCUDA2(points)
