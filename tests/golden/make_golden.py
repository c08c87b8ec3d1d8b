"""Generate the committed golden vectors.  Run in the BUILD container only:

    python tests/golden/make_golden.py

Part A (reference-derived, needs /root/reference): imports the two reference Python modules
that are importable here -- lib/utils/sh_utils.py and lib/utils/graphics_utils.py (torch/numpy
only) -- by file path and records inputs + outputs of

  * eval_sh(deg, sh, dirs)                 (sh_utils.py:57-112)  == forward.cu:20-71 minus +0.5/clamp
  * getWorld2View2, getProjectionMatrixK, getProjectionMatrix, fov2focal, focal2fov
                                           (graphics_utils.py:38-100)

These are the only parts of the path the reference itself can execute in this image (the
rasterizer proper is CUDA-only), so they are what pins the oracle.  Files: ref_sh.npz, ref_camera.npz;
plus IDFT (sh_utils.py:120-130) -> ref_idft.npz for the fused scene-graph composition.

Part B (oracle-generated regression vectors, NOT reference-derived -- "parity unpinned"):
small scenes run through oracle/gs_oracle.c; they catch regressions in either the oracle or the
HIP path and travel to the GPU box.  Files: oracle_<scene>.npz.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def part_a():
    sh_utils = _load(os.path.join(REF, "lib/utils/sh_utils.py"), "ref_sh_utils")
    gu = _load(os.path.join(REF, "lib/utils/graphics_utils.py"), "ref_graphics_utils")
    g = torch.Generator().manual_seed(1234)
    N = 96
    means = torch.randn(N, 3, generator=g) * 3.0
    means[:, 2] = means[:, 2].abs() + 1.0          # in front of an identity camera
    campos = torch.tensor([0.3, -0.2, -0.5])
    shs = torch.randn(N, 16, 3, generator=g) * 0.6
    dirs = means - campos[None]
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    out = {}
    for deg in range(4):
        M = (deg + 1) ** 2
        # reference layout: sh [..., C, (deg+1)^2]  (street_gaussian_renderer.py:183-185)
        res = sh_utils.eval_sh(deg, shs[:, :M, :].transpose(1, 2), dirs)
        out["eval_sh_deg%d" % deg] = res.numpy().astype(np.float32)
    np.savez(os.path.join(HERE, "ref_sh.npz"), means3D=means.numpy(), campos=campos.numpy(),
             shs=shs.numpy(), dirs=dirs.numpy(), **out)

    rng = np.random.RandomState(7)
    cams = {}
    for i in range(4):
        A = rng.randn(3, 3)
        Q, _ = np.linalg.qr(A)
        if np.linalg.det(Q) < 0:
            Q[:, 0] = -Q[:, 0]
        T = rng.randn(3) * 2.0
        cams["R%d" % i] = Q
        cams["T%d" % i] = T
        cams["w2v%d" % i] = gu.getWorld2View2(Q, T)
    K = np.array([[2083.09, 0.0, 960.0], [0.0, 2083.09, 640.0], [0.0, 0.0, 1.0]])
    cams["K"] = K
    cams["projK_1920x1280"] = gu.getProjectionMatrixK(K, 1280, 1920, 0.001, 1000.0).numpy()
    K2 = np.array([[700.0, 0.0, 300.5], [0.0, 650.0, 170.25], [0.0, 0.0, 1.0]])
    cams["K2"] = K2
    cams["projK_640x360"] = gu.getProjectionMatrixK(K2, 360, 640, 0.001, 1000.0).numpy()
    cams["proj_fov"] = gu.getProjectionMatrix(0.01, 100.0, 1.416, 0.506).numpy()
    cams["fov2focal"] = np.array([gu.fov2focal(1.416, 1242), gu.fov2focal(0.506, 375)])
    cams["focal2fov"] = np.array([gu.focal2fov(2083.09, 1920), gu.focal2fov(2083.09, 1280)])
    np.savez(os.path.join(HERE, "ref_camera.npz"), **cams)
    print("part A written (reference-derived)")


def part_a_idft():
    """IDFT(time, dim) of lib/utils/sh_utils.py:120-130 (the inverse-DFT weights of an actor's
    Fourier colour coefficients, gaussian_model_actor.py:73-82) -> ref_idft.npz.  Pins
    gaussianrpg_amd.composed.idft_weights and oracle/compose_torch.idft."""
    sh_utils = _load(os.path.join(REF, "lib/utils/sh_utils.py"), "ref_sh_utils_idft")
    times = [0.0, 0.1, 0.25, 0.5, 0.73, 1.0, 1.9]
    out = {}
    for d in (1, 2, 3, 5, 8):
        out["dim%d" % d] = np.stack([sh_utils.IDFT(t, d)[0].numpy() for t in times])
    np.savez(os.path.join(HERE, "ref_idft.npz"), times=np.array(times, np.float64), **out)


def part_a_sky():
    """Sky path: get_rays_torch (lib/utils/graphics_utils.py:186-207, importable) -> ref_rays.npz, and
    the face convention cube_to_dir (lib/models/sky_cubemap.py:139-146; the module itself needs
    nvdiffrast / cv2, so the function's source is extracted with ast and executed here, in the build
    container only -- the fixture holds inputs and outputs, no source text) -> ref_cube_dir.npz."""
    import ast
    gu = _load(os.path.join(REF, "lib/utils/graphics_utils.py"), "ref_gu_sky")
    g = torch.Generator().manual_seed(3)
    out = {}
    for n, (H, W) in enumerate([(5, 7), (12, 20)]):
        K = torch.tensor([[30.0 + n, 0.0, W / 2 + 0.3], [0.0, 28.0, H / 2 - 0.2], [0.0, 0.0, 1.0]])
        A = torch.randn(3, 3, generator=g)
        Q, _ = torch.linalg.qr(A)
        if torch.det(Q) < 0:
            Q[:, 0] = -Q[:, 0]
        T = torch.randn(3, generator=g)
        ro, rd = gu.get_rays_torch(H, W, K, Q, T, perturb=False)
        # train mode: perturb=True draws perturb_i, perturb_j = torch.rand(H, W) x2 from the global
        # generator (graphics_utils.py:194-196); the seed makes the planes reproducible
        torch.manual_seed(1000 + n)
        _, rdp = gu.get_rays_torch(H, W, K, Q, T, perturb=True)
        out.update({"rays_d_perturb%d" % n: rdp.numpy(), "perturb_seed%d" % n: np.array(1000 + n)})
        out.update({"K%d" % n: K.numpy(), "R%d" % n: Q.numpy(), "T%d" % n: T.numpy(),
                    "rays_o%d" % n: ro.numpy(), "rays_d%d" % n: rd.numpy(), "HW%d" % n: np.array([H, W])})
    np.savez(os.path.join(HERE, "ref_rays.npz"), **out)
    src = open(os.path.join(REF, "lib/models/sky_cubemap.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "cube_to_dir"][0]
    ns = {"torch": torch}
    exec(compile(ast.Module([fn], []), "cube_to_dir", "exec"), ns)
    gx, gy = torch.linspace(-0.9, 0.9, 7), torch.linspace(-0.8, 0.95, 6)
    yy, xx = torch.meshgrid(gy, gx, indexing='ij')
    dirs = np.stack([ns["cube_to_dir"](s, xx, yy).numpy() for s in range(6)])
    np.savez(os.path.join(HERE, "ref_cube_dir.npz"), x=xx.numpy(), y=yy.numpy(), dirs=dirs)


def _extract_functions(path, names):
    """Execute single functions of a reference module that cannot be imported as a whole here
    (its imports need roma / cv2 / CUDA): the named FunctionDefs are cut out with ast and executed
    in the BUILD container only; the fixtures hold inputs and outputs, never source text."""
    import ast
    src = open(path).read()
    fns = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(fns) == len(names), (names, [f.name for f in fns])
    ns = {"torch": torch, "np": np}
    exec(compile(ast.Module(fns, []), os.path.basename(path), "exec"), ns)
    return [ns[n] for n in names]


def part_a_quat():
    """Quaternion conventions (lib/utils/general_utils.py): quaternion_to_matrix_numpy (:103-122,
    numpy) and quaternion_raw_multiply (:220-238, torch, device-free) -> ref_quat.npz.  Pins
    (i) the (r, x, y, z) -> R convention of computeCov3D (CR/forward.cu:118-152) in gs_oracle.c
    and of the scene-graph composition, (ii) the Hamilton product obj_rot (x) q of
    street_gaussian_model.py:330-336 used by oracle/compose_torch.py and grpg_compose."""
    q2m, qmul = _extract_functions(os.path.join(REF, "lib/utils/general_utils.py"),
                                   ["quaternion_to_matrix_numpy", "quaternion_raw_multiply"])
    rng = np.random.RandomState(11)
    q = rng.randn(64, 4)
    q[0] = [1, 0, 0, 0]
    q[1] = [0, 1, 0, 0]
    q[2] = [0, 0, 1, 0]
    q[3] = [0, 0, 0, 1]
    q[4] = [np.cos(0.35), np.sin(0.35), 0, 0]       # rotation about x by 0.7 rad
    Rm = np.stack([q2m(qi.copy()) for qi in q])      # normalises internally
    a = torch.tensor(rng.randn(64, 4), dtype=torch.float32)
    b = torch.tensor(rng.randn(64, 4), dtype=torch.float32)
    prod = qmul(a, b).numpy()
    np.savez(os.path.join(HERE, "ref_quat.npz"), q=q, R=Rm, a=a.numpy(), b=b.numpy(), ab=prod)


def part_a_sh_bwd():
    """First reference-derived pin of the BACKWARD: torch.autograd through the reference's own
    eval_sh (lib/utils/sh_utils.py:57-112) and the direction normalisation its caller applies
    (street_gaussian_renderer.py:183-187: dir_pp / dir_pp.norm) -> ref_sh_bwd.npz.  The upstream
    gradient is masked where (eval_sh + 0.5) < 0, which is what the kernel's clamp does
    (CR/forward.cu:63-70, CR/backward.cu:36-39).  Checked against computeColorFromSH_bwd of
    gs_oracle.c (CR/backward.cu:20-139): dL_dsh and the direction part of dL_dmeans3D."""
    sh_utils = _load(os.path.join(REF, "lib/utils/sh_utils.py"), "ref_sh_utils_bwd")
    g = torch.Generator().manual_seed(4321)
    N = 128
    means = (torch.randn(N, 3, generator=g) * 3.0).double()
    campos = torch.tensor([0.3, -0.2, -0.5]).double()
    shs = (torch.randn(N, 16, 3, generator=g) * 1.5).double()
    gcol = torch.randn(N, 3, generator=g).double()
    out = dict(means3D=means.numpy().astype(np.float32), campos=campos.numpy().astype(np.float32),
               shs=shs.numpy().astype(np.float32), dL_dcolor=gcol.numpy().astype(np.float32))
    # float32-representable inputs, float64 autograd (exact reference gradient of those inputs)
    means = torch.tensor(out["means3D"]).double()
    campos = torch.tensor(out["campos"]).double()
    shs = torch.tensor(out["shs"]).double()
    gcol = torch.tensor(out["dL_dcolor"]).double()
    for deg in range(4):
        M = (deg + 1) ** 2
        m = means.clone().requires_grad_(True)
        s = shs[:, :M, :].clone().requires_grad_(True)
        d = m - campos[None]
        d = d / d.norm(dim=1, keepdim=True)
        res = sh_utils.eval_sh(deg, s.transpose(1, 2), d)
        mask = ((res.detach() + 0.5) >= 0).double()
        (res * gcol * mask).sum().backward()
        out["dL_dsh_deg%d" % deg] = s.grad.numpy()
        out["dL_dmeans_deg%d" % deg] = (m.grad.numpy() if m.grad is not None else np.zeros((N, 3)))
        out["clamped_deg%d" % deg] = (mask.numpy() == 0)
        out["margin_deg%d" % deg] = np.abs(res.detach().numpy() + 0.5)
    np.savez(os.path.join(HERE, "ref_sh_bwd.npz"), **out)


def part_a_cov3d():
    """The reference's PYTHON route to the 3D covariance -> ref_cov3d.npz: `get_covariance`
    (lib/models/gaussian_model.py:253) = build_covariance_from_scaling_rotation (:208-212) over
    build_scaling_rotation / quaternion_to_matrix / strip_symmetric (lib/utils/general_utils.py:278-287,
    :125-146, :88-100) -- what the renderer hands to the op as cov3D_precomp when
    `compute_cov3D_python` is set (gaussian_renderer.py:74-75), i.e. the reference's own statement of
    what computeCov3D (CR/forward.cu:118-152) computes from (scale, modifier, rotation), element
    order included.  The functions are cut out with ast and run here; their `device="cuda"` /
    `device='cuda'` keywords of torch.zeros are dropped (no CUDA in this container), the arithmetic
    is untouched."""
    import ast

    class TorchNoDevice:                      # torch, with torch.zeros(..., device=...) placed on the CPU
        def __getattr__(self, k):
            return getattr(torch, k)

        @staticmethod
        def zeros(*a, **kw):
            kw.pop("device", None)
            return torch.zeros(*a, **kw)
    ns = {"torch": TorchNoDevice(), "np": np}
    gu = ast.parse(open(os.path.join(REF, "lib/utils/general_utils.py")).read())
    names = ["quaternion_to_matrix", "build_scaling_rotation", "strip_lowerdiag", "strip_symmetric"]
    fns = [n for n in gu.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(fns) == len(names)
    exec(compile(ast.Module(fns, []), "general_utils.py", "exec"), ns)
    gm = ast.parse(open(os.path.join(REF, "lib/models/gaussian_model.py")).read())
    nested = [n for n in ast.walk(gm) if isinstance(n, ast.FunctionDef)
              and n.name == "build_covariance_from_scaling_rotation"]
    assert len(nested) == 1
    exec(compile(ast.Module(nested, []), "gaussian_model.py", "exec"), ns)
    cov_fn = ns["build_covariance_from_scaling_rotation"]
    g = torch.Generator().manual_seed(777)
    N = 96
    scales = torch.exp(-2.0 + 1.2 * torch.randn(N, 3, generator=g))
    scales[:8] *= torch.tensor([10.0, 0.1, 1.0])            # needles / pancakes
    q = torch.randn(N, 4, generator=g)
    q[0] = torch.tensor([1.0, 0, 0, 0])
    q[1] = torch.tensor([0.0, 0, 0, 1.0])
    rot = torch.nn.functional.normalize(q)                  # get_rotation (gaussian_model.py:229-230)
    out = dict(scales=scales.numpy(), rotations=rot.numpy())
    for name, mod in (("mod1", 1.0), ("mod06", 0.6)):
        out["cov_" + name] = cov_fn(scales, mod, rot).numpy()
    np.savez(os.path.join(HERE, "ref_cov3d.npz"), **out)


def part_a_ply_layout():
    """The trained-model file layouts, from the reference's own writer -> ref_ply_layout.npz:
    GaussianModel.make_ply / construct_list_of_attributes (lib/models/gaussian_model.py:80-96, 327-341)
    and GaussianModel.state_dict(is_final=True) (:182-205), cut out of the class with ast and run on a
    stand-in `self` that carries only the seven parameter tensors.  The fixture holds the tensors, the
    property names in file order and the element table make_ply returns; gaussianrpg_amd/checkpoint.py
    must turn that table back into the tensors (tests/test_model_checkpoint_loader.py)."""
    import ast
    import types
    cls = [n for n in ast.parse(open(os.path.join(REF, "lib/models/gaussian_model.py")).read()).body
           if isinstance(n, ast.ClassDef) and n.name == "GaussianModel"][0]
    names = ["make_ply", "construct_list_of_attributes", "state_dict"]
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(fns) == len(names)
    ns = {"torch": torch, "np": np}
    exec(compile(ast.Module(fns, []), "gaussian_model.py", "exec"), ns)
    g = torch.Generator().manual_seed(2024)
    N, S = 24, 3
    me = types.SimpleNamespace(
        _xyz=torch.randn(N, 3, generator=g) * 4, _features_dc=torch.randn(N, 1, 3, generator=g),
        _features_rest=torch.randn(N, 15, 3, generator=g) * 0.3, _scaling=torch.randn(N, 3, generator=g) - 2,
        _rotation=torch.randn(N, 4, generator=g), _opacity=torch.randn(N, 1, generator=g),
        _semantic=torch.rand(N, S, generator=g))
    me.construct_list_of_attributes = types.MethodType(ns["construct_list_of_attributes"], me)
    el = ns["make_ply"](me)
    sd = ns["state_dict"](me, True)
    np.savez(os.path.join(HERE, "ref_ply_layout.npz"),
             property_names=np.array(el.dtype.names), table=np.stack([el[k] for k in el.dtype.names], axis=1),
             state_dict_keys=np.array(list(sd.keys())),
             **{"sd_" + k: v.numpy() for k, v in sd.items()})


def part_a_compose():
    """The per-frame scene-graph composition AS A WHOLE, from the reference's own getters ->
    ref_compose.npz: StreetGaussianModel.get_xyz / get_rotation / get_scaling / get_opacity /
    get_features (lib/models/street_gaussian_model.py:296-453) over GaussianModel's activations
    (gaussian_model.py:224-251), GaussianModelActor.get_features_fourier (gaussian_model_actor.py:73-82),
    quaternion_to_matrix / quaternion_raw_multiply / matrix_to_quaternion (general_utils.py) and IDFT
    (sh_utils.py:120-130), including the training-time flip (street_gaussian_model.py:57-61,332,360).
    The methods are cut out of their classes with ast (decorators dropped) and run on stand-in objects
    that carry only tensors -- a background model and two actors with raw parameters, per-Gaussian
    obj_rots / obj_trans expanded the way parse_camera does (:262-281), a fixed flip mask.  No CUDA
    here: `device="cuda"` keywords of torch.zeros / torch.eye are dropped and Tensor.cuda() is the
    identity while this function runs; the arithmetic is the reference's."""
    import ast
    import types

    class TorchNoDevice:
        def __getattr__(self, k):
            return getattr(torch, k)

        @staticmethod
        def zeros(*a, **kw):
            kw.pop("device", None)
            return torch.zeros(*a, **kw)
    sh_utils = _load(os.path.join(REF, "lib/utils/sh_utils.py"), "ref_sh_utils_compose")
    ns = {"torch": TorchNoDevice(), "np": np, "IDFT": sh_utils.IDFT, "F": torch.nn.functional}

    def cut(path, cls_name, names, strip_decorators=True):
        tree = ast.parse(open(os.path.join(REF, path)).read())
        body = tree.body if cls_name is None else [n for n in tree.body if isinstance(n, ast.ClassDef)
                                                   and n.name == cls_name][0].body
        fns = [n for n in body if isinstance(n, ast.FunctionDef) and n.name in names]
        assert sorted(f.name for f in fns) == sorted(names), (names, [f.name for f in fns])
        for f in fns:
            if strip_decorators:
                f.decorator_list = []
        exec(compile(ast.Module(fns, []), os.path.basename(path), "exec"), ns)
        return [ns[n] for n in names]
    cut("lib/utils/general_utils.py", None,
        ["quaternion_to_matrix", "quaternion_raw_multiply", "matrix_to_quaternion", "_sqrt_positive_part"])
    g_xyz, g_rot, g_scal, g_opa, g_feat = cut(
        "lib/models/street_gaussian_model.py", "StreetGaussianModel",
        ["get_xyz", "get_rotation", "get_scaling", "get_opacity", "get_features"])
    (fourier,) = cut("lib/models/gaussian_model_actor.py", "GaussianModelActor", ["get_features_fourier"])

    g = torch.Generator().manual_seed(4242)
    r = lambda *shape: torch.randn(*shape, generator=g)      # noqa: E731

    def raw(n, fourier_dim):
        return dict(xyz=r(n, 3) * 3, scaling=r(n, 3) - 2, rotation=r(n, 4), opacity=r(n, 1),
                    features_dc=r(n, fourier_dim, 3), features_rest=r(n, 3, 3) * 0.2)

    def model_of(p, actor, frames=None):
        # GaussianModel's activated getters (gaussian_model.py:224-251): exp / normalize / sigmoid / cat
        m = types.SimpleNamespace(
            _features_dc=p["features_dc"], _features_rest=p["features_rest"], get_xyz=p["xyz"],
            get_scaling=torch.exp(p["scaling"]), get_rotation=torch.nn.functional.normalize(p["rotation"]),
            get_opacity=torch.sigmoid(p["opacity"]),
            get_features=torch.cat((p["features_dc"], p["features_rest"]), dim=1))
        if actor:
            m.start_frame, m.end_frame, m.fourier_scale = frames
            m.fourier_dim = p["features_dc"].shape[1]
            m.get_features_fourier = types.MethodType(fourier, m)
        return m
    params = {"background": raw(40, 1), "obj_003": raw(12, 5), "obj_011": raw(9, 5)}
    frames = {"obj_003": (10, 90, 1.0), "obj_011": (30, 50, 1.0)}
    frame = 37
    poses = {"obj_003": (torch.nn.functional.normalize(r(1, 4))[0], r(3) * 5),
             "obj_011": (torch.nn.functional.normalize(r(1, 4))[0], r(3) * 5)}
    old_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        flip_matrix = torch.eye(3).float() * -1        # street_gaussian_model.py:58-61
        flip_matrix[1, 1] = 1
        flip_matrix = ns["matrix_to_quaternion"](flip_matrix.unsqueeze(0))
        me = types.SimpleNamespace(get_visibility=lambda name: True, use_pose_correction=False,
                                   graph_obj_list=["obj_003", "obj_011"], frame=frame, flip_axis=1,
                                   flip_matrix=flip_matrix,
                                   background=model_of(params["background"], False))
        rots, trans, flips = [], [], []
        for name in me.graph_obj_list:
            m = model_of(params[name], True, frames[name])
            setattr(me, name, m)
            n = m.get_xyz.shape[0]
            rots.append(poses[name][0].expand(n, -1))                     # :275-276
            trans.append(poses[name][1].unsqueeze(0).expand(n, -1))
            flips.append(torch.rand(n, generator=g) < 0.5)                # :288-292 with flip_prob 0.5
        me.obj_rots, me.obj_trans, me.flip_mask = torch.cat(rots), torch.cat(trans), torch.cat(flips)
        out = dict(xyz=g_xyz(me), rotation=g_rot(me), scaling=g_scal(me), opacity=g_opa(me), features=g_feat(me))
    finally:
        torch.Tensor.cuda = old_cuda
    save = {"out_" + k: v.numpy() for k, v in out.items()}
    save["flip_matrix"] = flip_matrix.numpy()
    save["frame"] = frame
    for name, p in params.items():
        for k, v in p.items():
            save["%s.%s" % (name, k)] = v.numpy()
    for name in me.graph_obj_list:
        save[name + ".obj_rot"], save[name + ".obj_trans"] = poses[name][0].numpy(), poses[name][1].numpy()
        save[name + ".frames"] = np.array(frames[name], dtype=np.float64)
    save["obj_003.flip"], save["obj_011.flip"] = flips[0].numpy(), flips[1].numpy()
    np.savez(os.path.join(HERE, "ref_compose.npz"), **save)


def part_a_binding_trace():
    """The Python -> _C boundary of the operator, traced under the reference's own wrapper ->
    ref_binding_trace.json: submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py
    is imported with a RECORDING stand-in for its compiled `_C` (tests/helpers.py BindingRecorder) and
    driven through a training forward + backward (SH + scales/rotations, and colors_precomp +
    cov3D_precomp), markVisible and visible_filter.  The fixture holds, per `_C` call, the entry point's
    name and which caller-side object or plain value sits in every argument position, what the wrapper
    returns in which order, and which of `_C`'s nine gradients reaches which input."""
    import json
    import types
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import BindingRecorder, binding_trace
    rec = BindingRecorder()
    pkg = types.ModuleType("ref_dgr")
    pkg.__path__ = []
    pkg._C = rec
    sys.modules["ref_dgr"] = pkg
    sys.modules["ref_dgr._C"] = rec
    path = os.path.join(REF, "submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py")
    mod = types.ModuleType("ref_dgr.wrapper")
    mod.__package__ = "ref_dgr"
    exec(compile(open(path).read(), path, "exec"), mod.__dict__)
    trace = binding_trace(mod, rec)
    json.dump(trace, open(os.path.join(HERE, "ref_binding_trace.json"), "w"), indent=1)


def scenes():
    """(name, scene, camera, extra kwargs) of the oracle-generated regression fixtures."""
    from gaussianrpg_amd import harness as hz
    g = torch.Generator().manual_seed(99)
    out = []
    out.append(("toy_deg1", hz.toy_scene(1500, seed=1, sh_degree=1),
                hz.trajectory_camera(0, W=96, H=64), dict(bg=[0.0, 0.0, 0.0])))
    out.append(("toy_deg3_sem", hz.toy_scene(1200, seed=2, sh_degree=3),
                hz.trajectory_camera(0, W=80, H=50), dict(bg=[0.2, 0.4, 0.6], S=3)))
    out.append(("smoke_deg0", hz.smoke_scene(600, seed=0), hz.smoke_camera(64, 48),
                dict(bg=[0.0, 0.0, 0.0])))
    out.append(("street_small", hz.street_scene(6000, seed=5), hz.trajectory_camera(2, W=160, H=112),
                dict(bg=[1.0, 1.0, 1.0])))
    return out, g


def part_b():
    sys.path.insert(0, ROOT)
    import oracle
    from gaussianrpg_amd import harness as hz
    sc_list, g = scenes()
    for name, sc, cam, extra in sc_list:
        P = sc.means3D.shape[0]
        S = extra.get("S", 0)
        sem = torch.rand(P, S, generator=g) if S else None
        bg = torch.tensor(extra["bg"], dtype=torch.float32)
        kw = {k: v for k, v in hz.settings_kwargs(cam, sc.sh_degree, bg=bg).items()
              if k not in ("prefiltered", "debug")}
        o = oracle.forward(sc.means3D, sc.opacity, shs=sc.shs, scales=sc.scales,
                           rotations=sc.rotations, semantics=sem, **kw)
        H, W = cam.image_height, cam.image_width
        gc = torch.randn(3, H, W, generator=g).numpy()
        gd = (0.1 * torch.randn(1, H, W, generator=g)).numpy()
        ga = torch.randn(1, H, W, generator=g).numpy()
        gs = torch.randn(S, H, W, generator=g).numpy()
        gr = oracle.backward(o, gc, gd, ga, gs)
        np.savez_compressed(
            os.path.join(HERE, "oracle_%s.npz" % name),
            means3D=sc.means3D.numpy(), opacity=sc.opacity.numpy(), scales=sc.scales.numpy(),
            rotations=sc.rotations.numpy(), shs=sc.shs.numpy(),
            semantics=(sem.numpy() if S else np.zeros((P, 0), np.float32)),
            sh_degree=sc.sh_degree, bg=bg.numpy(), H=H, W=W, tanfovx=cam.tanfovx,
            tanfovy=cam.tanfovy, viewmatrix=cam.viewmatrix.numpy(),
            projmatrix=cam.projmatrix.numpy(), campos=cam.campos.numpy(),
            radii=o["radii"], num_rendered=o["num_rendered"], keys_sorted=o["keys_sorted"],
            point_list=o["point_list"], ranges=o["ranges"], n_contrib=o["n_contrib"],
            fragile=o["fragile"], color=o["color"], depth=o["depth"], alpha=o["alpha"],
            semantic=o["semantic"], means2D=o["means2D"], depths=o["depths"],
            conic_opacity=o["conic_opacity"], rgb=o["rgb"],
            grad_color=gc, grad_depth=gd, grad_alpha=ga, grad_semantic=gs,
            **{k: v for k, v in gr.items()})
        print("wrote oracle_%s.npz  P=%d V=%d R=%d" % (name, P, int((o["radii"] > 0).sum()),
                                                      o["num_rendered"]))


if __name__ == "__main__":
    if os.path.isdir(REF):
        part_a()
        part_a_idft()
        part_a_sky()
        part_a_quat()
        part_a_sh_bwd()
        part_a_cov3d()
        part_a_ply_layout()
        part_a_compose()
        part_a_binding_trace()
    else:
        print("no /root/reference here: skipping part A (reference-derived vectors)")
    part_b()
