"""-m gpu: a producer / consumer hand-over of the render that times out is an ERROR, not a silently
wrong image (csrc/render_fwd.hip pc_fail, csrc/api.hip check_async_error; the reference's loud
failure policy: cuda_rasterizer/auxiliary.h:166-173).

The timeout cannot be provoked in the product build (the spin limit is 2^24 polls), so the test
loads a VARIANT of the C-ABI library -- the same sources, render_fwd.hip compiled with
-DGRPG_PC_SPIN_LIMIT=0 (gaussianrpg_amd/build.py VARIANTS), in which every wait that does not
succeed at its first poll gives up -- through ctypes, next to the product library."""
import ctypes
import os

import pytest
import torch

from gaussianrpg_amd import build as gbuild
from gaussianrpg_amd import harness as hz

pytestmark = pytest.mark.gpu

ALLOC = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)


def _forward(lib, dev, sc, cam, debug, keep):
    d = sc.to(dev)
    P, Hh, Ww = d.means3D.shape[0], cam.image_height, cam.image_width

    def alloc(nbytes, _user):
        t = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        keep.append(t)
        return t.data_ptr()
    cb = ALLOC(alloc)
    f32 = lambda t: ctypes.c_void_p(t.data_ptr())       # noqa: E731
    bg = torch.zeros(3, device=dev)
    view, proj, campos = cam.viewmatrix.to(dev), cam.projmatrix.to(dev), cam.campos.to(dev)
    color = torch.empty(3, Hh, Ww, device=dev)
    depth = torch.empty(1, Hh, Ww, device=dev)
    alpha = torch.empty(1, Hh, Ww, device=dev)
    radii = torch.empty(P, dtype=torch.int32, device=dev)
    keep += [d, bg, view, proj, campos, color, depth, alpha, radii, cb]
    lib.grpg_forward.restype = ctypes.c_int
    lib.grpg_last_error.restype = ctypes.c_char_p
    stream = torch.cuda.current_stream().cuda_stream
    R = lib.grpg_forward(cb, None, cb, None, cb, None, P, 1, 4, 0, f32(bg), Ww, Hh, f32(d.means3D),
                         f32(d.shs), None, None, f32(d.opacity), f32(d.scales), ctypes.c_float(1.0),
                         f32(d.rotations), None, f32(view), f32(proj), f32(campos),
                         ctypes.c_float(cam.tanfovx), ctypes.c_float(cam.tanfovy), 0, f32(color),
                         f32(depth), f32(alpha), None, ctypes.c_void_p(radii.data_ptr()),
                         1 if debug else 0, ctypes.c_void_p(stream))
    torch.cuda.synchronize()
    return R, lib.grpg_last_error().decode()


def test_pc_timeout_is_reported():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    dev = torch.device("cuda:0")
    path = gbuild.variant_path("pcspin0")
    if not os.path.exists(path):
        gbuild.build_variant("pcspin0")      # normally built by __graft_entry__.build() and shipped
    lib = ctypes.CDLL(path)
    # tiles with 10-18 k entries: rendered by producer / consumer wave pairs (>= 8192 entries)
    long_sc = hz.toy_scene(40000, seed=21, sh_degree=1, depth=6.0, spread=0.8, scale=0.015)
    long_cam = hz.trajectory_camera(0, W=64, H=64)
    short_sc, short_cam = hz.toy_scene(3000, seed=4, sh_degree=1), hz.trajectory_camera(0, W=200, H=136)
    keep = []
    # a frame without such tiles is untouched by the variant
    R, err = _forward(lib, dev, short_sc, short_cam, False, keep)
    assert R > 0 and err == ""
    # non-debug: the frame that suffers the timeout returns normally (nobody waits for its end) ...
    R, err = _forward(lib, dev, long_sc, long_cam, False, keep)
    assert R > 0, err
    # ... and the NEXT entry point of the thread reports it, once
    R2, err2 = _forward(lib, dev, short_sc, short_cam, False, keep)
    assert R2 == -3 and "hand-over timed out" in err2, (R2, err2)     # GRPG_ERR_HIP
    R3, err3 = _forward(lib, dev, short_sc, short_cam, False, keep)
    assert R3 > 0 and err3 == ""
    # debug = true: the forward that suffered it fails itself
    R4, err4 = _forward(lib, dev, long_sc, long_cam, True, keep)
    assert R4 == -3 and "hand-over timed out" in err4, (R4, err4)
    R5, err5 = _forward(lib, dev, short_sc, short_cam, False, keep)
    assert R5 > 0 and err5 == ""
