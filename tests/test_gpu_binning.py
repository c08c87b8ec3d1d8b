"""GPU tests (-m gpu) of the two tile-binning algorithms behind grpg_forward: the hierarchical path
(coarse super-tile partition + per-tile count + direct fill, csrc/hier_binning.hip) and the sort
path (emit + stable partition, csrc/binning.hip) must produce the SAME sorted keys, point list, tile
ranges and images, bit for bit -- on the first frame of a shape (exact capacity), on later frames
(speculative capacity) and at grid sizes that exercise every branch of the coarse partition:
one tile (no coarse pass), <= 256 super-tiles (one pass, runs from the digit totals) and > 256
super-tiles (two passes + run detection on masked keys).  One case is also pinned to the oracle.
"""
import numpy as np
import pytest
import torch

import oracle
from gaussianrpg_amd import harness as hz
from helpers import oracle_kwargs

pytestmark = pytest.mark.gpu

SORT, HIER = 0, 1


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X (no ROCm device visible)")
    return torch.device("cuda:0")


@pytest.fixture()
def restore_algorithm():
    from gaussianrpg_amd.rasterizer import _C
    before = _C.get_binning_algorithm()
    yield _C
    _C.set_binning_algorithm(before)
    _C.reset_capacity_hints()


def _forward(dev, sc, cam):
    from gaussianrpg_amd.rasterizer import _C, debug_export
    camd = hz.CameraTensors(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy,
                            cam.viewmatrix.to(dev), cam.projmatrix.to(dev), cam.campos.to(dev))
    kw = hz.settings_kwargs(camd, sc.sh_degree, bg=torch.zeros(3, device=dev))
    d = sc.to(dev)
    P = d.means3D.shape[0]
    e = torch.Tensor([])
    args = (kw["bg"], d.means3D, e, torch.zeros(P, 0, device=dev), d.opacity, d.scales, d.rotations,
            kw["scale_modifier"], e, kw["viewmatrix"], kw["projmatrix"], kw["tanfovx"], kw["tanfovy"],
            kw["image_height"], kw["image_width"], d.shs, kw["sh_degree"], kw["campos"],
            kw["prefiltered"], kw["debug"])
    R, color, depth, alpha, _sem, radii, geom, binning, img = _C.rasterize_gaussians(*args)
    dbg = debug_export(geom, binning, img, P, R, kw["image_height"], kw["image_width"])
    torch.cuda.synchronize()
    out = dict(R=R, color=color, depth=depth, alpha=alpha, radii=radii, **dbg)
    return {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in out.items()}


EXACT = ("R", "radii", "tiles_touched", "keys_sorted", "point_list", "ranges", "n_contrib", "color",
         "depth", "alpha")


@pytest.mark.parametrize("W,H,P,depth", [
    (16, 16, 300, 6.0),          # one tile, one super-tile: no coarse pass at all
    (256, 256, 4000, 4.0),       # 4 super-tiles
    (1920, 1280, 20000, 3.0),    # 150 super-tiles: one coarse pass, runs from the digit totals
    (4096, 2304, 20000, 3.0),    # 576 super-tiles: two coarse passes, runs found on the sorted keys
    (1000, 200, 3000, 2.0),      # ragged grid: partial tiles and partial super-tiles on both axes
])
def test_hierarchical_equals_sort_binning(dev, restore_algorithm, W, H, P, depth):
    _C = restore_algorithm
    sc = hz.toy_scene(P, seed=W + P, sh_degree=1, depth=depth)
    cam = hz.trajectory_camera(1, W=W, H=H)
    got = {}
    for alg in (SORT, HIER):
        _C.set_binning_algorithm(alg)
        _C.reset_capacity_hints()
        frames = [_forward(dev, sc, cam) for _ in range(3)]   # exact, then two speculative frames
        for f in frames[1:]:
            for k in EXACT:
                np.testing.assert_array_equal(f[k], frames[0][k], err_msg="%s frame-to-frame (alg %d)" % (k, alg))
        got[alg] = frames[0]
    assert got[HIER]["R"] > 0
    for k in EXACT:
        np.testing.assert_array_equal(got[HIER][k], got[SORT][k], err_msg=k)


def test_hierarchical_binning_matches_oracle_on_a_ragged_grid(dev, restore_algorithm):
    _C = restore_algorithm
    _C.set_binning_algorithm(HIER)
    _C.reset_capacity_hints()
    W, H = 200, 136     # 13 x 9 tiles: 2 x 2 super-tiles, the last ones partial
    sc = hz.toy_scene(1500, seed=77, sh_degree=1, depth=3.0)
    cam = hz.trajectory_camera(2, W=W, H=H)
    o = oracle.forward(sc.means3D, sc.opacity, shs=sc.shs, scales=sc.scales, rotations=sc.rotations,
                       **oracle_kwargs(cam, sc.sh_degree))
    for _ in range(2):   # exact-capacity frame, then a speculative one
        got = _forward(dev, sc, cam)
        assert got["R"] == o["num_rendered"]
        np.testing.assert_array_equal(got["keys_sorted"].view(np.uint64), o["keys_sorted"])
        np.testing.assert_array_equal(got["point_list"].view(np.uint32), o["point_list"])
        np.testing.assert_array_equal(got["ranges"].view(np.uint32), o["ranges"])


def test_growing_scene_overflows_the_remembered_capacity(dev, restore_algorithm):
    """Same (P, W, H) key, but the second scene covers far more tiles than the first: the
    speculative capacities (instances AND coarse pairs) overflow and the frame's tail is redone."""
    _C = restore_algorithm
    W, H, P = 640, 384, 6000
    cam = hz.trajectory_camera(0, W=W, H=H)
    small = hz.toy_scene(P, seed=5, sh_degree=1, depth=30.0)     # far away: ~1 tile per Gaussian
    big = hz.toy_scene(P, seed=5, sh_degree=1, depth=1.5)        # close: many tiles each
    ref = {}
    for alg in (SORT, HIER):
        _C.set_binning_algorithm(alg)
        _C.reset_capacity_hints()
        a = _forward(dev, small, cam)
        b = _forward(dev, big, cam)
        assert b["R"] > 2 * a["R"]
        ref[alg] = b
    for k in EXACT:
        np.testing.assert_array_equal(ref[HIER][k], ref[SORT][k], err_msg=k)
