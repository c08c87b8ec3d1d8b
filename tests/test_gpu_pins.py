"""-m gpu: the HIP path against the REFERENCE-DERIVED fixtures, directly -- not through the oracle.

tests/test_reference_pins.py and tests/test_oracle_golden.py check the CPU oracle against vectors
produced by executing the reference's own Python (tests/golden/make_golden.py); the GPU parity
tests check HIP == oracle.  Here the HIP kernels themselves meet the same vectors, called through
the drop-in Python API / the _C binding:

* ref_sh.npz      eval_sh, degrees 0-3 (R/lib/utils/sh_utils.py:57-112)      -> preprocess' SH->RGB
* ref_sh_bwd.npz  float64 autograd through eval_sh + direction normalisation  -> preprocess backward's
                  dL_dsh (per-Gaussian basis: dL_dsh / dL_dcolor), through _C.rasterize_gaussians_backward
* ref_quat.npz    quaternion_to_matrix_numpy (R/lib/utils/general_utils.py:103-122) -> the (r,x,y,z)
                  convention of computeCov3D as seen in the conic the kernel stores
* ref_cov3d.npz   build_covariance_from_scaling_rotation (R/lib/models/gaussian_model.py:208-212) -> the
                  op's internal computeCov3D (scale modifier, entry order) as seen in radii and conics
* ref_camera.npz  getWorld2View2 / getProjectionMatrixK (R/lib/utils/graphics_utils.py:38-94) ->
                  means2D / depths of preprocess for matrices built exactly like the reference's Camera
"""
import os

import numpy as np
import pytest
import torch

from gaussianrpg_amd import harness as hz
from helpers import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X (no ROCm device visible)")
    return torch.device("cuda:0")


def _camera_seeing_all(means, W=4096, H=4096, f=300.0):
    """A view (pure translation) that puts the whole cloud in front of a wide camera: the SH direction
    depends on `campos` alone (forward.cu:25-27), which the tests set to the fixture's, so the view is
    free -- it only has to make every Gaussian visible."""
    T = np.array([-means[:, 0].mean(), -means[:, 1].mean(), 6.0 - means[:, 2].min()])
    return hz.make_camera(np.eye(3), T, W=W, H=H, fx=f, fy=f, cx=W / 2, cy=H / 2)


def _forward_raw(dev, cam, sh_degree, means, opacity, scales, rotations, shs=None, colors=None,
                 campos=None, requires_grad=False):
    """_C.rasterize_gaussians + debug_export; returns (outputs, blobs, decoded intermediates)."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    from gaussianrpg_amd.rasterizer import _C, debug_export
    camd = hz.CameraTensors(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy,
                            cam.viewmatrix.to(dev), cam.projmatrix.to(dev),
                            (cam.campos if campos is None else torch.as_tensor(campos)).float().to(dev))
    rs = GaussianRasterizationSettings(**hz.settings_kwargs(camd, sh_degree, bg=torch.zeros(3, device=dev)))
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(dev)   # noqa: E731
    e = torch.Tensor([])
    P = means.shape[0]
    args = (rs.bg, t(means), e if colors is None else t(colors), torch.zeros(P, 0, device=dev), t(opacity),
            t(scales), t(rotations), 1.0, e, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy,
            rs.image_height, rs.image_width, e if shs is None else t(shs), rs.sh_degree, rs.campos, False, False)
    out = _C.rasterize_gaussians(*args)
    R, color, depth, alpha, semantic, radii, geom, binning, img = out
    dbg = debug_export(geom, binning, img, P, R, rs.image_height, rs.image_width)
    torch.cuda.synchronize()
    return rs, args, out, {k: v.cpu().numpy() for k, v in dbg.items()}


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_hip_sh_to_rgb_matches_reference_eval_sh(dev, deg):
    z = np.load(os.path.join(GOLDEN, "ref_sh.npz"))
    means, campos, shs = z["means3D"], z["campos"], z["shs"]
    M = (deg + 1) ** 2
    P = means.shape[0]
    cam = _camera_seeing_all(means)
    _, _, out, dbg = _forward_raw(dev, cam, deg, means, np.full((P, 1), 0.5, np.float32),
                                  np.full((P, 3), 0.05, np.float32),
                                  np.tile(np.array([[1, 0, 0, 0]], np.float32), (P, 1)),
                                  shs=shs[:, :M, :].copy(), campos=campos)
    vis = out[5].cpu().numpy() > 0
    assert vis.all()
    ref = z["eval_sh_deg%d" % deg] + 0.5          # forward.cu:60; clamped at 0 (:62-66)
    np.testing.assert_allclose(dbg["rgb"][vis], np.maximum(ref, 0.0)[vis], rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_hip_sh_backward_matches_reference_autograd(dev, deg):
    """dL_dsh[i, k, c] = basis_k(dir_i) * dL_dcolor[i, c] for an unclamped channel (backward.cu:36-139):
    the per-Gaussian basis the kernel applied, recovered as dL_dsh / dL_dcolor from the arrays
    _C.rasterize_gaussians_backward returns (the reference's binding returns both,
    rasterize_points.cu:166-176,219), must equal the basis float64 autograd applied in the fixture."""
    from gaussianrpg_amd.rasterizer import _C
    z = np.load(os.path.join(GOLDEN, "ref_sh_bwd.npz"))
    means, campos = z["means3D"], z["campos"]
    M = (deg + 1) ** 2
    shs = z["shs"][:, :M, :].copy()
    P = means.shape[0]
    cam = _camera_seeing_all(means)
    rs, args, out, _ = _forward_raw(dev, cam, deg, means, np.full((P, 1), 0.6, np.float32),
                                    np.full((P, 3), 0.03, np.float32),
                                    np.tile(np.array([[1, 0, 0, 0]], np.float32), (P, 1)),
                                    shs=shs, campos=campos)
    R, color, depth, alpha, semantic, radii, geom, binning, img = out
    g = torch.Generator().manual_seed(deg)
    H, W = rs.image_height, rs.image_width
    # a random positive image gradient: every visible Gaussian gets a non-zero colour gradient
    g_color = (torch.rand(3, H, W, generator=g) + 0.5).to(dev)
    zero1 = torch.zeros(1, H, W, device=dev)
    grads = _C.rasterize_gaussians_backward(
        args[0], args[1], radii, args[2], args[5], args[6], 1.0, args[8], args[9], args[10], rs.tanfovx,
        rs.tanfovy, g_color, zero1, zero1, torch.zeros(0, H, W, device=dev), args[15], rs.sh_degree,
        rs.campos, geom, R, binning, img, alpha, args[3], False)
    dL_dcolors, dL_dsh = grads[1].cpu().numpy().astype(np.float64), grads[5].cpu().numpy().astype(np.float64)
    assert dL_dcolors.shape == (P, 3) and grads[4].shape == (P, 6)      # the reference's shapes
    vis = radii.cpu().numpy() > 0
    assert vis.all()
    ref_sh, ref_c = z["dL_dsh_deg%d" % deg], z["dL_dcolor"].astype(np.float64)
    clamped = z["clamped_deg%d" % deg]
    margin = z["margin_deg%d" % deg]          # |colour before the clamp|: stay away from the kink
    checked = 0
    for c in range(3):
        ok = vis & ~clamped[:, c] & (np.abs(dL_dcolors[:, c]) > 1e-12) & (np.abs(ref_c[:, c]) > 1e-6) & \
            (margin[:, c] > 1e-4)
        basis_hip = dL_dsh[ok, :, c] / dL_dcolors[ok, c][:, None]
        basis_ref = ref_sh[ok, :, c] / ref_c[ok, c][:, None]
        np.testing.assert_allclose(basis_hip, basis_ref, rtol=2e-5, atol=2e-6)
        checked += int(ok.sum())
        # a clamped channel passes no gradient on (backward.cu:31-34)
        cl = vis & clamped[:, c] & (margin[:, c] > 1e-4)
        assert not dL_dsh[cl, :, c].any()
    assert checked > 200


def test_hip_conic_uses_reference_quaternion_convention(dev):
    """The conic preprocess stores for anisotropic Gaussians == EWA of Sigma = R diag(s^2) R^T with the
    REFERENCE's rotation matrices (float64); the transposed matrices give visibly different conics."""
    z = np.load(os.path.join(GOLDEN, "ref_quat.npz"))
    q = (z["q"] / np.linalg.norm(z["q"], axis=1, keepdims=True)).astype(np.float32)
    n = q.shape[0]
    rng = np.random.RandomState(5)
    means = np.c_[rng.uniform(-1, 1, n), rng.uniform(-0.6, 0.6, n), rng.uniform(4, 8, n)].astype(np.float32)
    scales = (np.c_[np.full(n, 0.30), np.full(n, 0.05), np.full(n, 0.11)] *
              rng.uniform(0.8, 1.25, (n, 1))).astype(np.float32)
    W, H, f = 256, 192, 120.0
    cam = hz.make_camera(np.eye(3), np.zeros(3), W=W, H=H, fx=f, fy=f, cx=W / 2, cy=H / 2)
    _, _, out, dbg = _forward_raw(dev, cam, 0, means, np.full((n, 1), 0.5, np.float32), scales, q,
                                  colors=np.ones((n, 3), np.float32))
    assert (out[5].cpu().numpy() > 0).all()

    def conic(Rm):
        S2 = scales.astype(np.float64) ** 2
        Sig = np.einsum("nij,nj,nkj->nik", Rm, S2, Rm)
        t = means.astype(np.float64)                       # identity view: t = mean
        J = np.zeros((n, 2, 3))
        J[:, 0, 0] = f / t[:, 2]; J[:, 0, 2] = -f * t[:, 0] / t[:, 2] ** 2
        J[:, 1, 1] = f / t[:, 2]; J[:, 1, 2] = -f * t[:, 1] / t[:, 2] ** 2
        cov = np.einsum("nij,njk,nlk->nil", J, Sig, J)
        a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3      # forward.cu:108-110
        det = a * c - b * b
        return np.stack([c / det, -b / det, a / det], axis=1)
    ref = conic(z["R"])
    np.testing.assert_allclose(dbg["conic_opacity"][:, :3], ref, rtol=2e-4, atol=1e-7)
    wrong = conic(np.transpose(z["R"], (0, 2, 1)))
    assert (np.abs(wrong - ref).max(axis=1) > 1e-3 * np.abs(ref).max(axis=1)).mean() > 0.5


@pytest.mark.parametrize("mod,key", [(1.0, "cov_mod1"), (0.6, "cov_mod06")])
def test_hip_covariance_matches_reference_python_covariance(dev, mod, key):
    """ref_cov3d.npz: what preprocess derives from (scales, scale_modifier, rotations) == what it derives
    from cov3D_precomp = the reference's OWN Python covariance of the same Gaussians
    (get_covariance, gaussian_model.py:208-212,253; the `compute_cov3D_python` route of
    gaussian_renderer.py:74-75): same radii, same conics (float32 rounding of two routes to the same
    matrix), so the op's internal computeCov3D and the order of the six entries are the reference's."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    from gaussianrpg_amd.rasterizer import _C, debug_export
    z = np.load(os.path.join(GOLDEN, "ref_cov3d.npz"))
    scales, rots, cov = z["scales"], z["rotations"], z[key]
    n = scales.shape[0]
    rng = np.random.RandomState(8)
    means = np.c_[rng.uniform(-1, 1, n), rng.uniform(-0.6, 0.6, n), rng.uniform(6, 9, n)].astype(np.float32)
    cam = hz.make_camera(np.eye(3), np.zeros(3), W=512, H=384, fx=200.0, fy=200.0, cx=256, cy=192)
    camd = hz.CameraTensors(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy,
                            cam.viewmatrix.to(dev), cam.projmatrix.to(dev), cam.campos.to(dev))
    rs = GaussianRasterizationSettings(**hz.settings_kwargs(camd, 0, bg=torch.zeros(3, device=dev),
                                                            scale_modifier=mod))
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(dev)   # noqa: E731
    e = torch.Tensor([])

    def run(use_cov):
        args = (rs.bg, t(means), t(np.ones((n, 3), np.float32)), torch.zeros(n, 0, device=dev),
                t(np.full((n, 1), 0.5, np.float32)), e if use_cov else t(scales), e if use_cov else t(rots),
                rs.scale_modifier, t(cov) if use_cov else e, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
                rs.tanfovy, rs.image_height, rs.image_width, e, 0, rs.campos, False, False)
        out = _C.rasterize_gaussians(*args)
        dbg = debug_export(out[6], out[7], out[8], n, out[0], rs.image_height, rs.image_width)
        torch.cuda.synchronize()
        return out[5].cpu().numpy(), dbg["conic_opacity"].cpu().numpy()
    radii_sr, conic_sr = run(False)
    radii_cov, conic_cov = run(True)
    assert (radii_sr > 0).all() and (radii_cov > 0).all()
    # the radius is ceil(3 sqrt(lambda_max)): one unit of slack where the two roundings straddle an integer
    assert np.abs(radii_sr - radii_cov).max() <= 1 and (radii_sr == radii_cov).mean() >= 0.95
    np.testing.assert_allclose(conic_sr, conic_cov, rtol=2e-4, atol=1e-7)
    # and the entry order matters: cov3D with xy / xz swapped gives other conics
    cov_swapped = cov[:, [0, 2, 1, 3, 4, 5]]
    args = (rs.bg, t(means), t(np.ones((n, 3), np.float32)), torch.zeros(n, 0, device=dev),
            t(np.full((n, 1), 0.5, np.float32)), e, e, rs.scale_modifier, t(cov_swapped), rs.viewmatrix,
            rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, e, 0, rs.campos, False, False)
    out = _C.rasterize_gaussians(*args)
    wrong = debug_export(out[6], out[7], out[8], n, out[0], rs.image_height, rs.image_width)["conic_opacity"].cpu().numpy()
    vis = out[5].cpu().numpy() > 0
    assert (np.abs(wrong[vis, :3] - conic_sr[vis, :3]).max(axis=1) > 1e-3 * np.abs(conic_sr[vis, :3]).max(axis=1)).mean() > 0.5


def test_hip_projection_uses_reference_matrices(dev):
    """means2D / depths of preprocess == (Proj @ W2C) applied in float64 with the reference's OWN
    getWorld2View2 / getProjectionMatrixK outputs, handed to the op the way the reference's Camera
    does (transposed, camera_utils.py:50-58); ndc2Pix: ((v + 1) S - 1) / 2."""
    z = np.load(os.path.join(GOLDEN, "ref_camera.npz"))
    W, H = 640, 360
    K2 = z["K2"]
    w2c32, proj32 = z["w2v1"], z["projK_640x360"]
    view_t = torch.tensor(w2c32).transpose(0, 1).contiguous()
    full_t = (view_t.unsqueeze(0).bmm(torch.tensor(proj32).transpose(0, 1).unsqueeze(0))).squeeze(0)
    campos = view_t.inverse()[3, :3]
    cam = hz.CameraTensors(H, W, W / (2 * K2[0, 0]), H / (2 * K2[1, 1]), view_t, full_t.contiguous(), campos)
    w2c = w2c32.astype(np.float64)
    full = proj32.astype(np.float64) @ w2c
    rng = np.random.RandomState(3)
    pc = rng.randn(400, 3) * np.array([1.0, 0.6, 1.0]) + np.array([0, 0, 5.0])   # camera space
    pw = (np.linalg.inv(w2c) @ np.c_[pc, np.ones(400)].T).T[:, :3].astype(np.float32)
    _, _, out, dbg = _forward_raw(dev, cam, 0, pw, np.full((400, 1), 0.5, np.float32),
                                  np.full((400, 3), 0.02, np.float32),
                                  np.tile(np.array([[1, 0, 0, 0]], np.float32), (400, 1)),
                                  colors=np.ones((400, 3), np.float32))
    hom = (full @ np.c_[pw.astype(np.float64), np.ones(400)].T).T
    ndc = hom[:, :2] / hom[:, 3:4]
    pix = ((ndc + 1.0) * np.array([W, H]) - 1.0) * 0.5
    vis = out[5].cpu().numpy() > 0
    assert vis.sum() > 100
    np.testing.assert_allclose(dbg["means2D"][vis], pix[vis], rtol=0, atol=2e-2)
    np.testing.assert_allclose(dbg["depths"][vis], (w2c @ np.c_[pw, np.ones(400)].T).T[vis, 2], rtol=1e-5)
