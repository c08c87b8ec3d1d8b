"""-m gpu: grpg_backward through ctypes (no _C, no torch types in the call): the backward WRITES every
element of every gradient array it is handed -- nothing may arrive zero-filled, nothing may stay
uninitialised -- and refuses the calls it cannot serve instead of dereferencing NULL or writing out
of bounds (include/grpg_rasterizer.h; the reference accumulates into eleven zero-filled arrays,
DGR/rasterize_points.cu:166-176).

Every output buffer is poisoned with NaN before the call.
"""
import ctypes

import numpy as np
import pytest
import torch

from gaussianrpg_amd import harness as hz
from gaussianrpg_amd.build import LIB_PATH

pytestmark = pytest.mark.gpu

ALLOC = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)
GRPG_ERR_INVALID_ARGUMENT, GRPG_ERR_BAD_BUFFER = -1, -5
GRPG_FORWARD_NO_BACKWARD = 1


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X (no ROCm device visible)")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def lib():
    lb = ctypes.CDLL(LIB_PATH)
    lb.grpg_forward.restype = ctypes.c_int
    lb.grpg_forward_flags.restype = ctypes.c_int
    lb.grpg_backward.restype = ctypes.c_int
    lb.grpg_last_error.restype = ctypes.c_char_p
    return lb


def p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class Frame:
    """One forward through the C ABI; keeps everything the backward needs alive."""

    def __init__(self, lib, dev, sc, cam, colors=None, cov=None, flags=0):
        self.lib, self.dev = lib, dev
        d = sc.to(dev)
        self.P, self.H, self.W = d.means3D.shape[0], cam.image_height, cam.image_width
        self.M = 0 if colors is not None else d.shs.shape[1]
        self.D = sc.sh_degree
        self.blobs = []

        def alloc(nbytes, _user):
            t = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
            self.blobs.append(t)
            return t.data_ptr()
        self.cb = ALLOC(alloc)
        self.bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
        self.view, self.proj, self.campos = cam.viewmatrix.to(dev), cam.projmatrix.to(dev), cam.campos.to(dev)
        self.tanx, self.tany = ctypes.c_float(cam.tanfovx), ctypes.c_float(cam.tanfovy)
        self.means, self.opac = d.means3D, d.opacity
        self.shs = None if colors is not None else d.shs
        self.colors = None if colors is None else colors.to(dev)
        self.scales = None if cov is not None else d.scales
        self.rots = None if cov is not None else d.rotations
        self.cov = None if cov is None else cov.to(dev)
        self.color = torch.empty(3, self.H, self.W, device=dev)
        self.depth = torch.empty(1, self.H, self.W, device=dev)
        self.alpha = torch.empty(1, self.H, self.W, device=dev)
        self.radii = torch.empty(self.P, dtype=torch.int32, device=dev)
        self.stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        common = (self.cb, None, self.cb, None, self.cb, None, self.P, self.D, self.M, 0, p(self.bg), self.W,
                  self.H, p(self.means), p(self.shs), p(self.colors), None, p(self.opac), p(self.scales),
                  ctypes.c_float(1.0), p(self.rots), p(self.cov), p(self.view), p(self.proj), p(self.campos),
                  self.tanx, self.tany, 0, p(self.color), p(self.depth), p(self.alpha), None, p(self.radii),
                  0, self.stream)
        self.R = lib.grpg_forward_flags(*common, flags) if flags else lib.grpg_forward(*common)
        assert self.R >= 0, lib.grpg_last_error()
        torch.cuda.synchronize()
        # allocation order of the forward: geometry, image, binning (re-carved last after an overflow)
        self.geom, self.image, self.binning = self.blobs[0], self.blobs[1], self.blobs[-1]

    def backward(self, outs, g_color=None):
        """outs: dict name -> tensor or None for the eleven gradient arrays of grpg_backward."""
        gen = torch.Generator().manual_seed(3)
        gc = (torch.randn(3, self.H, self.W, generator=gen) if g_color is None else g_color).to(self.dev)
        gd = torch.randn(1, self.H, self.W, generator=gen).to(self.dev)
        ga = torch.randn(1, self.H, self.W, generator=gen).to(self.dev)
        rc = self.lib.grpg_backward(
            self.P, self.D, self.M, self.R, 0, p(self.bg), self.W, self.H, p(self.means), p(self.shs),
            p(self.colors), None, p(self.alpha), p(self.scales), ctypes.c_float(1.0), p(self.rots), p(self.cov),
            p(self.view), p(self.proj), p(self.campos), self.tanx, self.tany, p(self.radii), p(self.geom),
            p(self.binning), p(self.image), p(gc), p(gd), p(ga), None,
            p(outs["mean2D"]), p(outs["conic"]), p(outs["opacity"]), p(outs["color"]), p(outs["depth"]),
            p(outs["mean3D"]), p(outs["cov3D"]), p(outs["sh"]), p(outs["scale"]), p(outs["rot"]), None,
            0, self.stream)
        torch.cuda.synchronize()
        return rc, self.lib.grpg_last_error().decode()


def poisoned(dev, P, M, names):
    shapes = dict(mean2D=(P, 3), conic=(P, 4), opacity=(P, 1), color=(P, 3), depth=(P,), mean3D=(P, 3),
                  cov3D=(P, 6), sh=(P, max(M, 1), 3), scale=(P, 3), rot=(P, 4))
    return {k: (torch.full(shapes[k], float("nan"), device=dev) if k in names else None) for k in shapes}


def test_every_gradient_element_is_written_incl_culled(dev, lib):
    sc = hz.toy_scene(3000, seed=4, sh_degree=1)
    means = sc.means3D.clone()
    means[::7, 2] = -3.0            # every 7th Gaussian behind the camera: culled
    sc = sc._replace(means3D=means)
    fr = Frame(lib, dev, sc, hz.trajectory_camera(0, W=200, H=136))
    # all ten arrays, the optional intermediates included (the reference's binding returns them all)
    outs = poisoned(dev, fr.P, fr.M, ("mean2D", "conic", "opacity", "color", "depth", "mean3D", "cov3D", "sh",
                                      "scale", "rot"))
    rc, err = fr.backward(outs)
    assert rc == 0, err
    culled = (fr.radii == 0)
    assert int(culled.sum()) >= fr.P // 7
    for k, t in outs.items():
        assert bool(torch.isfinite(t).all()), "%s holds unwritten (NaN) elements" % k
        assert float(t[culled].abs().max()) == 0.0, "%s: a culled Gaussian has a gradient" % k
    assert float(outs["mean3D"].abs().max()) > 0 and float(outs["sh"].abs().max()) > 0
    # only the arrays the caller reads: the others may be NULL
    outs2 = poisoned(dev, fr.P, fr.M, ("mean2D", "opacity", "mean3D", "sh", "scale", "rot"))
    rc, err = fr.backward(outs2)
    assert rc == 0, err
    for k in ("mean2D", "opacity", "mean3D", "sh", "scale", "rot"):
        assert bool(torch.isfinite(outs2[k]).all()), k
        # float atomics reorder the sums of the blend backward: equal to rounding, not bitwise
        assert torch.allclose(outs2[k], outs[k], rtol=1e-3, atol=1e-5 * float(outs[k].abs().max()))


def test_nothing_rendered_gives_written_zeros(dev, lib):
    sc = hz.toy_scene(500, seed=5, sh_degree=1)
    means = sc.means3D.clone()
    means[:, 2] = -2.0              # everything behind the camera: R == 0
    fr = Frame(lib, dev, sc._replace(means3D=means), hz.trajectory_camera(0, W=96, H=64))
    assert fr.R == 0
    outs = poisoned(dev, fr.P, fr.M, ("mean2D", "opacity", "mean3D", "sh", "scale", "rot"))
    rc, err = fr.backward(outs)
    assert rc == 0, err
    for k in ("mean2D", "opacity", "mean3D", "sh", "scale", "rot"):
        assert float(outs[k].abs().max()) == 0.0, k      # NaN would fail this too


def test_precomputed_colour_and_covariance_path(dev, lib):
    import oracle
    from helpers import oracle_kwargs
    sc, cam = hz.toy_scene(1200, seed=13, sh_degree=1), hz.trajectory_camera(0, W=160, H=96)
    colors = torch.rand(1200, 3, generator=torch.Generator().manual_seed(2))
    o0 = oracle.forward(sc.means3D, sc.opacity, colors_precomp=colors, scales=sc.scales, rotations=sc.rotations,
                        render=False, **oracle_kwargs(cam, 0))
    fr = Frame(lib, dev, sc._replace(sh_degree=0), cam, colors=colors, cov=torch.tensor(o0["cov3D"]))
    outs = poisoned(dev, fr.P, fr.M, ("mean2D", "opacity", "mean3D", "color", "cov3D"))
    rc, err = fr.backward(outs)
    assert rc == 0, err
    for k in ("mean2D", "opacity", "mean3D", "color", "cov3D"):
        assert bool(torch.isfinite(outs[k]).all()), k
    assert float(outs["color"].abs().max()) > 0 and float(outs["cov3D"].abs().max()) > 0
    # the gradient array of an input that IS in use must be there
    outs_bad = poisoned(dev, fr.P, fr.M, ("mean2D", "opacity", "mean3D", "color"))
    rc, err = fr.backward(outs_bad)
    assert rc == GRPG_ERR_INVALID_ARGUMENT and "cov3D_precomp" in err


def test_missing_gradient_arrays_are_refused(dev, lib):
    fr = Frame(lib, dev, hz.toy_scene(800, seed=6, sh_degree=1), hz.trajectory_camera(0, W=96, H=64))
    for missing, word in (("sh", "dL_dsh"), ("scale", "dL_dscale"), ("rot", "dL_drot"), ("mean3D", "NULL")):
        names = {"mean2D", "opacity", "mean3D", "sh", "scale", "rot"} - {missing}
        rc, err = fr.backward(poisoned(dev, fr.P, fr.M, names))
        assert rc == GRPG_ERR_INVALID_ARGUMENT and word in err, (missing, rc, err)


def test_backward_refuses_an_evaluation_forwards_blob(dev, lib):
    """An evaluation forward (GRPG_FORWARD_NO_BACKWARD) carves the geometry blob WITHOUT the backward's
    gradient records: pairing it with grpg_backward must be an error return, not an out-of-bounds
    device write."""
    sc, cam = hz.toy_scene(800, seed=6, sh_degree=1), hz.trajectory_camera(0, W=96, H=64)
    fr = Frame(lib, dev, sc, cam, flags=GRPG_FORWARD_NO_BACKWARD)
    rc, err = fr.backward(poisoned(dev, fr.P, fr.M, ("mean2D", "opacity", "mean3D", "sh", "scale", "rot")))
    assert rc == GRPG_ERR_BAD_BUFFER and "evaluation forward" in err, (rc, err)
    # the training forward of the same shape is fine
    fr2 = Frame(lib, dev, sc, cam)
    rc, err = fr2.backward(poisoned(dev, fr2.P, fr2.M, ("mean2D", "opacity", "mean3D", "sh", "scale", "rot")))
    assert rc == 0, err


def test_blob_guard_is_settled_by_the_header_not_by_the_address(dev, lib):
    """the library's table of blob addresses is advisory.  A training blob COPIED to an
    address the library has never seen is served (its header says it has the gradient records); a
    copied evaluation blob at an unknown address is still refused (the header says no), and so is
    memory that holds no geometry blob at all."""
    sc, cam = hz.toy_scene(800, seed=6, sh_degree=1), hz.trajectory_camera(0, W=96, H=64)
    names = ("mean2D", "opacity", "mean3D", "sh", "scale", "rot")
    ref = Frame(lib, dev, sc, cam)
    want = poisoned(dev, ref.P, ref.M, names)
    rc, err = ref.backward(want)
    assert rc == 0, err
    # (a) training blob moved to a fresh address
    fr = Frame(lib, dev, sc, cam)
    fr.geom = fr.geom.clone()
    got = poisoned(dev, fr.P, fr.M, names)
    rc, err = fr.backward(got)
    assert rc == 0, err
    for k in ("opacity", "scale"):     # float atomics: equal to rounding
        assert torch.allclose(got[k], want[k], rtol=1e-3, atol=1e-6 * float(want[k].abs().max())), k
    # (b) evaluation blob copied to an unknown address: refused by its header
    ev3 = Frame(lib, dev, sc, cam, flags=GRPG_FORWARD_NO_BACKWARD)
    ev3.geom = torch.cat([ev3.geom, torch.zeros(ev3.P * 64 + 4096, dtype=torch.uint8, device=dev)])
    rc, err = ev3.backward(poisoned(dev, ev3.P, ev3.M, names))
    assert rc == GRPG_ERR_BAD_BUFFER and "evaluation forward" in err, (rc, err)
    # (c) something that is not a geometry blob at all
    ev3.geom = torch.zeros_like(ev3.geom)
    rc, err = ev3.backward(poisoned(dev, ev3.P, ev3.M, names))
    assert rc == GRPG_ERR_BAD_BUFFER and "magic" in err, (rc, err)


def test_allocation_failures_are_reported_and_leave_the_library_usable(dev, lib):
    """A grpg_alloc_fn that returns NULL -- for the geometry blob, the image blob or the binning blob
    (the last one is asked for in the middle of the frame, behind launches that are already enqueued) --
    makes grpg_forward return GRPG_ERR_ALLOC with a message, never a crash; the next call with a working
    allocator renders the same frame as if nothing had happened."""
    GRPG_ERR_ALLOC = -4
    sc, cam = hz.toy_scene(2500, seed=17, sh_degree=1), hz.trajectory_camera(0, W=160, H=112)
    good = Frame(lib, dev, sc, cam)
    want = good.color.clone()
    d = sc.to(dev)
    P, H, W = d.means3D.shape[0], cam.image_height, cam.image_width
    view, proj, campos = cam.viewmatrix.to(dev), cam.projmatrix.to(dev), cam.campos.to(dev)
    bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
    for fail_at in (1, 2, 3):
        blobs, calls = [], [0]

        def alloc(nbytes, _user):
            calls[0] += 1
            if calls[0] == fail_at:
                return 0
            t = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
            blobs.append(t)
            return t.data_ptr()
        cb = ALLOC(alloc)
        color = torch.empty(3, H, W, device=dev)
        depth, alpha = torch.empty(1, H, W, device=dev), torch.empty(1, H, W, device=dev)
        radii = torch.empty(P, dtype=torch.int32, device=dev)
        rc = lib.grpg_forward(cb, None, cb, None, cb, None, P, sc.sh_degree, d.shs.shape[1], 0, p(bg), W, H,
                              p(d.means3D), p(d.shs), None, None, p(d.opacity), p(d.scales), ctypes.c_float(1.0),
                              p(d.rotations), None, p(view), p(proj), p(campos), ctypes.c_float(cam.tanfovx),
                              ctypes.c_float(cam.tanfovy), 0, p(color), p(depth), p(alpha), None, p(radii), 0,
                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert rc == GRPG_ERR_ALLOC, (fail_at, rc, lib.grpg_last_error())
        assert b"allocation failed" in lib.grpg_last_error()
        again = Frame(lib, dev, sc, cam)
        assert again.R == good.R and torch.equal(again.color, want)
