"""CPU oracle for the rasterizer hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker.  The product path
(``gaussianrpg_amd`` / ``diff_gaussian_rasterization``) never does.

``oracle.forward`` / ``oracle.backward`` drive ``gs_oracle.c`` (the scalar C
restatement of cuda_rasterizer/{forward,backward,rasterizer_impl}.cu) through
ctypes.  See the header of ``gs_oracle.c`` for the parity-pinning status
("parity unpinned" for the CUDA stages; SH and camera conventions pinned by
``tests/golden``).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgs_oracle.so")
_LIB_OMP_PATH = os.path.join(_HERE, "libgs_oracle_omp.so")
_lib = None
_use_omp = False

BLOCK_X = 16
BLOCK_Y = 16


def build(force=False):
    """Compile gs_oracle.c -> libgs_oracle.so (scalar) and libgs_oracle_omp.so (OpenMP over
    Gaussians / pixel rows, same results) with gcc (see Makefile)."""
    src = os.path.join(_HERE, "gs_oracle.c")
    for path, extra in ((_LIB_PATH, []), (_LIB_OMP_PATH, ["-fopenmp"])):
        if (not force and os.path.exists(path) and os.path.getmtime(path) >= os.path.getmtime(src)):
            continue
        subprocess.check_call(
            ["gcc", "-O2", "-fPIC", "-std=c99", "-ffp-contract=off", "-fno-fast-math"] + extra +
            ["-shared", "-o", path, src, "-lm"])
    return _LIB_PATH


def use_openmp(flag=True):
    """Select the OpenMP build of the C oracle for subsequent calls (bench.py's cpu_baseline leg;
    results are identical to the scalar build, tests/test_oracle_golden.py checks it)."""
    global _lib, _use_omp
    if bool(flag) != _use_omp:
        _use_omp = bool(flag)
        _lib = None


def num_threads():
    lib = _load()
    lib.gso_num_threads.restype = ctypes.c_int
    return int(lib.gso_num_threads())


def _load():
    global _lib
    if _lib is None:
        path = _LIB_OMP_PATH if _use_omp else _LIB_PATH
        src = os.path.join(_HERE, "gs_oracle.c")
        if not os.path.exists(path) or (os.path.exists(src)
                                        and os.path.getmtime(path) < os.path.getmtime(src)):
            build()
        _lib = ctypes.CDLL(path)
        _lib.gso_preprocess.restype = ctypes.c_int64
        _lib.gso_higher_msb.restype = ctypes.c_uint32
    return _lib


def _np(x, dtype=np.float32):
    if x is None:
        return None
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(x, dtype=dtype))


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f(x):
    return ctypes.c_float(float(x))


def higher_msb(n):
    return int(_load().gso_higher_msb(ctypes.c_uint32(n)))


def forward(means3D, opacities, *, image_height, image_width, tanfovx, tanfovy, bg,
            scale_modifier, viewmatrix, projmatrix, sh_degree, campos, shs=None,
            colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
            semantics=None, render=True):
    """Full forward of the reference algorithm.  Returns a dict of numpy arrays with
    every intermediate the reference keeps in its three blobs (CR/rasterizer_impl.cu:155-193)
    plus the five outputs."""
    lib = _load()
    means3D = _np(means3D).reshape(-1, 3)
    P = means3D.shape[0]
    H, W = int(image_height), int(image_width)
    opacities = _np(opacities).reshape(-1)
    shs = _np(shs)
    colors_precomp = _np(colors_precomp)
    scales = _np(scales)
    rotations = _np(rotations)
    cov3D_precomp = _np(cov3D_precomp)
    semantics = _np(semantics)
    if shs is not None and shs.size == 0:
        shs = None
    if colors_precomp is not None and colors_precomp.size == 0:
        colors_precomp = None
    if scales is not None and scales.size == 0:
        scales = None
    if rotations is not None and rotations.size == 0:
        rotations = None
    if cov3D_precomp is not None and cov3D_precomp.size == 0:
        cov3D_precomp = None
    S = 0 if semantics is None else int(semantics.reshape(P, -1).shape[1])
    M = 0 if shs is None else int(shs.shape[1])
    view = _np(viewmatrix).reshape(16)
    proj = _np(projmatrix).reshape(16)
    campos = _np(campos).reshape(3)
    bg = _np(bg).reshape(3)
    o = dict(P=P, W=W, H=H, S=S, M=M)
    o["radii"] = np.zeros(P, np.int32)
    o["means2D"] = np.zeros((P, 2), np.float32)
    o["depths"] = np.zeros(P, np.float32)
    o["cov3D"] = np.zeros((P, 6), np.float32)
    o["rgb"] = np.zeros((P, 3), np.float32)
    o["conic_opacity"] = np.zeros((P, 4), np.float32)
    o["clamped"] = np.zeros((P, 3), np.uint8)
    o["tiles_touched"] = np.zeros(P, np.uint32)
    o["point_offsets"] = np.zeros(P, np.uint32)
    R = lib.gso_preprocess(
        P, int(sh_degree), M, _p(means3D), _p(scales), _f(scale_modifier), _p(rotations),
        _p(opacities), _p(shs), _p(cov3D_precomp), _p(colors_precomp), _p(view), _p(proj),
        _p(campos), W, H, _f(tanfovx), _f(tanfovy), _p(o["radii"]), _p(o["means2D"]),
        _p(o["depths"]), _p(o["cov3D"]), _p(o["rgb"]), _p(o["conic_opacity"]), _p(o["clamped"]),
        _p(o["tiles_touched"]), _p(o["point_offsets"])) if P > 0 else 0
    R = int(R)
    o["num_rendered"] = R
    gx, gy = (W + BLOCK_X - 1) // BLOCK_X, (H + BLOCK_Y - 1) // BLOCK_Y
    o["grid"] = (gx, gy)
    o["keys_sorted"] = np.zeros(R, np.uint64)
    o["point_list"] = np.zeros(R, np.uint32)
    o["ranges"] = np.zeros((gx * gy, 2), np.uint32)
    if P > 0:
        lib.gso_bin(P, ctypes.c_int64(R), W, H, _p(o["radii"]), _p(o["means2D"]), _p(o["depths"]),
                    _p(o["point_offsets"]), _p(o["keys_sorted"]), _p(o["point_list"]), _p(o["ranges"]))
    o["color"] = np.zeros((3, H, W), np.float32)
    o["depth"] = np.zeros((1, H, W), np.float32)
    o["alpha"] = np.zeros((1, H, W), np.float32)
    o["semantic"] = np.zeros((S, H, W), np.float32)
    o["n_contrib"] = np.zeros((H, W), np.uint32)
    o["fragile"] = np.zeros((H, W), np.uint8)
    o["features"] = colors_precomp if colors_precomp is not None else o["rgb"]
    # P == 0: the binding launches nothing and returns zero-filled outputs
    # (rasterize_points.cu:85-86,123) -- colour stays black, not background.
    if render and P > 0:
        lib.gso_render(W, H, S, _p(o["ranges"]), _p(o["point_list"]), _p(o["means2D"]),
                       _p(o["features"]), _p(o["depths"]), _p(semantics), _p(o["conic_opacity"]),
                       _p(bg), _p(o["color"]), _p(o["depth"]), _p(o["alpha"]), _p(o["semantic"]),
                       _p(o["n_contrib"]), _p(o["fragile"]))
    o["_inputs"] = dict(means3D=means3D, opacities=opacities, shs=shs, colors_precomp=colors_precomp,
                        scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp,
                        semantics=semantics, view=view, proj=proj, campos=campos, bg=bg,
                        tanfovx=float(tanfovx), tanfovy=float(tanfovy),
                        scale_modifier=float(scale_modifier), sh_degree=int(sh_degree))
    return o


def backward(fwd, grad_color, grad_depth, grad_alpha, grad_semantic=None, blend_f64=False):
    """Backward of the reference algorithm given a forward() result and the four
    pixel-plane gradients.  Returns the 9 gradients the binding returns
    (rasterize_points.cu:219) plus the internal dL_dconic / dL_ddepths.

    blend_f64=True: the ARBITER (gs_oracle.c gso_render_backward_f64) -- the blend stage's gradient in float64
    from the exact final transmittance, same accept decisions; the preprocess backward behind it is the same
    float32 restatement.  For settling HIP-vs-oracle disagreements at sizes float64 autograd does not reach."""
    lib = _load()
    i = fwd["_inputs"]
    P, W, H, S, M = fwd["P"], fwd["W"], fwd["H"], fwd["S"], fwd["M"]
    gc = _np(grad_color).reshape(3, H, W)
    gd = _np(grad_depth).reshape(H, W)
    ga = _np(grad_alpha).reshape(H, W)
    gs = _np(grad_semantic).reshape(S, H, W) if S > 0 else np.zeros((0, H, W), np.float32)
    g = {}
    g["dL_dmeans2D"] = np.zeros((P, 3), np.float32)
    g["dL_dconic"] = np.zeros((P, 4), np.float32)
    g["dL_dopacity"] = np.zeros((P, 1), np.float32)
    g["dL_dcolors"] = np.zeros((P, 3), np.float32)
    g["dL_ddepths"] = np.zeros((P, 1), np.float32)
    g["dL_dsemantic"] = np.zeros((P, S), np.float32)
    g["dL_dmeans3D"] = np.zeros((P, 3), np.float32)
    g["dL_dcov3D"] = np.zeros((P, 6), np.float32)
    g["dL_dsh"] = np.zeros((P, M, 3), np.float32)
    g["dL_dscales"] = np.zeros((P, 3), np.float32)
    g["dL_drotations"] = np.zeros((P, 4), np.float32)
    if P == 0:
        return g
    if blend_f64:
        lib.gso_render_backward_f64(
            P, W, H, S, _p(fwd["ranges"]), _p(fwd["point_list"]), _p(i["bg"]), _p(fwd["means2D"]),
            _p(fwd["conic_opacity"]), _p(fwd["features"]), _p(fwd["depths"]), _p(i["semantics"]),
            _p(fwd["n_contrib"]), _p(gc), _p(gd), _p(ga), _p(gs),
            _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dopacity"]), _p(g["dL_dcolors"]),
            _p(g["dL_ddepths"]), _p(g["dL_dsemantic"]))
    else:
        lib.gso_render_backward(
            P, W, H, S, _p(fwd["ranges"]), _p(fwd["point_list"]), _p(i["bg"]), _p(fwd["means2D"]),
            _p(fwd["conic_opacity"]), _p(fwd["features"]), _p(fwd["depths"]), _p(i["semantics"]),
            _p(fwd["alpha"]), _p(fwd["n_contrib"]), _p(gc), _p(gd), _p(ga), _p(gs),
            _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dopacity"]), _p(g["dL_dcolors"]),
            _p(g["dL_ddepths"]), _p(g["dL_dsemantic"]))
    cov3Ds = i["cov3D_precomp"] if i["cov3D_precomp"] is not None else fwd["cov3D"]
    lib.gso_preprocess_backward(
        P, i["sh_degree"], M, _p(i["means3D"]), _p(fwd["radii"]), _p(i["shs"]), _p(fwd["clamped"]),
        _p(i["scales"]), _p(i["rotations"]), _f(i["scale_modifier"]), _p(cov3Ds), _p(i["view"]),
        _p(i["proj"]), W, H, _f(i["tanfovx"]), _f(i["tanfovy"]), _p(i["campos"]),
        _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dmeans3D"]), _p(g["dL_dcolors"]),
        _p(g["dL_ddepths"]), _p(g["dL_dcov3D"]), _p(g["dL_dsh"]), _p(g["dL_dscales"]),
        _p(g["dL_drotations"]))
    return g


def mark_visible(means3D, viewmatrix, projmatrix):
    lib = _load()
    means3D = _np(means3D).reshape(-1, 3)
    P = means3D.shape[0]
    out = np.zeros(P, np.uint8)
    if P:
        lib.gso_mark_visible(P, _p(means3D), _p(_np(viewmatrix).reshape(16)),
                             _p(_np(projmatrix).reshape(16)), _p(out))
    return out.astype(bool)
