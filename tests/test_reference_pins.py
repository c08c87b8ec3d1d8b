"""CPU tests (-m "not gpu"): reference-derived pins added in round 3.

The fixtures were generated in the build container by executing the reference's own Python
(tests/golden/make_golden.py part_a_quat / part_a_sh_bwd):

* ref_quat.npz   -- quaternion_to_matrix_numpy (R/lib/utils/general_utils.py:103-122) and
                    quaternion_raw_multiply (:220-238);
* ref_sh_bwd.npz -- torch.autograd through eval_sh (R/lib/utils/sh_utils.py:57-112) + the caller's
                    direction normalisation: the first reference-derived pin of the backward.

* ref_cov3d.npz  -- (round 4) build_covariance_from_scaling_rotation (R/lib/models/gaussian_model.py:
                    208-212) over build_scaling_rotation / quaternion_to_matrix / strip_symmetric.

They pin: gs_oracle.c's quat_to_R_glm / computeCov3D convention (CR/forward.cu:118-152),
oracle/compose_torch.py's quaternion_to_matrix and quaternion_raw_multiply, and gs_oracle.c's
computeColorFromSH_bwd (CR/backward.cu:20-139: dL_dsh and the direction part of dL_dmeans3D).
The matching GPU checks (HIP kernels against the same fixtures) live in tests/test_gpu_pins.py.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import compose_torch as ct
from gaussianrpg_amd import harness as hz
from helpers import GOLDEN, oracle_kwargs


def _quat_fixture():
    z = np.load(os.path.join(GOLDEN, "ref_quat.npz"))
    return {k: z[k] for k in z.files}


def cov3d_from_reference_R(Rm, scales):
    """Sigma = R diag(s^2) R^T in float64 from the reference's rotation matrices; the six unique
    entries in the order the kernel stores them (CR/forward.cu:146-151)."""
    S2 = scales.astype(np.float64) ** 2
    Sig = np.einsum("nij,nj,nkj->nik", Rm, S2, Rm)
    return np.stack([Sig[:, 0, 0], Sig[:, 0, 1], Sig[:, 0, 2], Sig[:, 1, 1], Sig[:, 1, 2],
                     Sig[:, 2, 2]], axis=1)


def quat_scene(fx):
    """Gaussians in front of an identity camera carrying the fixture's (normalised) quaternions
    and strongly anisotropic scales, so that a transposed R or a permuted (r,x,y,z) shows."""
    q = fx["q"] / np.linalg.norm(fx["q"], axis=1, keepdims=True)
    n = q.shape[0]
    rng = np.random.RandomState(5)
    means = np.c_[rng.uniform(-1, 1, n), rng.uniform(-0.6, 0.6, n), rng.uniform(4, 8, n)].astype(np.float32)
    scales = np.c_[np.full(n, 0.30), np.full(n, 0.05), np.full(n, 0.11)].astype(np.float32)
    scales *= rng.uniform(0.8, 1.25, (n, 1)).astype(np.float32)
    return means, scales, q.astype(np.float32)


def test_cov3d_uses_reference_quaternion_convention():
    fx = _quat_fixture()
    means, scales, q = quat_scene(fx)
    n = q.shape[0]
    cam = hz.make_camera(np.eye(3), np.zeros(3), W=256, H=192, fx=120.0, fy=120.0, cx=128, cy=96)
    o = oracle.forward(means, np.full((n, 1), 0.5, np.float32), colors_precomp=np.ones((n, 3), np.float32),
                       scales=scales, rotations=q, render=False, **oracle_kwargs(cam, 0))
    vis = o["radii"] > 0
    assert vis.all()
    ref = cov3d_from_reference_R(fx["R"], scales)
    np.testing.assert_allclose(o["cov3D"], ref, rtol=2e-5, atol=2e-7)
    # the convention is observable: the transposed matrix gives a different covariance
    wrong = cov3d_from_reference_R(np.transpose(fx["R"], (0, 2, 1)), scales)
    assert np.abs(wrong - ref).max() > 1e-3


@pytest.mark.parametrize("mod,key", [(1.0, "cov_mod1"), (0.6, "cov_mod06")])
def test_cov3d_matches_reference_python_covariance(mod, key):
    """ref_cov3d.npz (round 4): the reference's own Python statement of computeCov3D --
    build_covariance_from_scaling_rotation(scaling, modifier, rotation) (gaussian_model.py:208-212,
    the cov3D_precomp route of gaussian_renderer.py:74-75), element order of strip_symmetric -- against
    gs_oracle.c's computeCov3D (CR/forward.cu:118-152) on the same scales, rotations and modifier."""
    z = np.load(os.path.join(GOLDEN, "ref_cov3d.npz"))
    scales, rots = z["scales"], z["rotations"]
    n = scales.shape[0]
    rng = np.random.RandomState(8)
    means = np.c_[rng.uniform(-1, 1, n), rng.uniform(-0.6, 0.6, n), rng.uniform(6, 9, n)].astype(np.float32)
    cam = hz.make_camera(np.eye(3), np.zeros(3), W=512, H=384, fx=200.0, fy=200.0, cx=256, cy=192)
    o = oracle.forward(means, np.full((n, 1), 0.5, np.float32), colors_precomp=np.ones((n, 3), np.float32),
                       scales=scales, rotations=rots, render=False,
                       **oracle_kwargs(cam, 0, scale_modifier=mod))
    assert (o["radii"] > 0).all()
    scale = np.abs(z[key]).max(axis=1, keepdims=True)       # per Gaussian: entries span many decades
    np.testing.assert_allclose(o["cov3D"] / scale, z[key] / scale, rtol=0, atol=3e-6)
    # the order of the six entries is observable (xy <-> xz swapped is another matrix)
    swapped = z[key][:, [0, 2, 1, 3, 4, 5]]
    assert np.abs(swapped / scale - o["cov3D"] / scale).max() > 1e-2


def test_compose_restatement_quaternions_match_reference():
    fx = _quat_fixture()
    R = ct.quaternion_to_matrix(torch.tensor(fx["q"], dtype=torch.float32)).numpy()
    np.testing.assert_allclose(R, fx["R"], rtol=0, atol=3e-6)
    ab = ct.quaternion_raw_multiply(torch.tensor(fx["a"]), torch.tensor(fx["b"])).numpy()
    np.testing.assert_array_equal(ab, fx["ab"])


def sh_bwd_oracle(z, deg):
    """computeColorFromSH_bwd alone: gso_preprocess_backward with zero dL_dmean2D / dL_dconic /
    dL_ddepth and no scales, so dL_dmeans is the SH direction term only."""
    lib = oracle._load()
    means = np.ascontiguousarray(z["means3D"], np.float32)
    P = means.shape[0]
    M = (deg + 1) ** 2
    shs = np.ascontiguousarray(z["shs"][:, :M, :], np.float32)
    clamped = np.ascontiguousarray(z["clamped_deg%d" % deg].astype(np.uint8))
    radii = np.ones(P, np.int32)
    zeros = lambda *s: np.zeros(s, np.float32)
    view = np.eye(4, dtype=np.float32).reshape(16)
    # a benign projection (w = z): the mean2D chain is multiplied by zero gradients anyway
    proj = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0], np.float32)
    dmeans, dcov, dsh = zeros(P, 3), zeros(P, 6), zeros(P, M, 3)
    dscale, drot = zeros(P, 3), zeros(P, 4)
    dcol = np.ascontiguousarray(z["dL_dcolor"], np.float32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    f = lambda v: ctypes.c_float(v)
    lib.gso_preprocess_backward(
        P, deg, M, p(means), p(radii), p(shs), p(clamped), None, None, f(1.0), p(zeros(P, 6)),
        p(view), p(proj), 64, 64, f(1.0), f(1.0), p(np.ascontiguousarray(z["campos"], np.float32)),
        p(zeros(P, 3)), p(zeros(P, 4)), p(dmeans), p(dcol), p(zeros(P)), p(dcov), p(dsh),
        p(dscale), p(drot))
    return dsh, dmeans


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_backward_matches_reference_autograd(deg):
    z = np.load(os.path.join(GOLDEN, "ref_sh_bwd.npz"))
    dsh, dmeans = sh_bwd_oracle(z, deg)
    ref_sh, ref_m = z["dL_dsh_deg%d" % deg], z["dL_dmeans_deg%d" % deg]
    assert np.isfinite(dmeans).all()
    np.testing.assert_allclose(dsh, ref_sh, rtol=2e-5, atol=2e-6)
    scale = np.abs(ref_m).max() + 1e-12
    np.testing.assert_allclose(dmeans, ref_m, rtol=1e-4, atol=2e-5 * scale)
    if deg > 0:
        assert np.abs(ref_m).max() > 1e-3      # the direction term is really exercised
    # clamped channels receive no gradient (CR/backward.cu:36-39)
    cl = z["clamped_deg%d" % deg]
    assert cl.any() and not dsh[:, 0, :][cl].any()


def test_composition_restatement_matches_the_references_own_getters():
    """ref_compose.npz (round 4): the per-frame composition AS A WHOLE -- StreetGaussianModel's getters
    run in the reference's code on a background model and two posed, partly flipped actors
    (street_gaussian_model.py:296-453, gaussian_model_actor.py:73-82) -- against
    oracle/compose_torch.py, the checker of the fused composition (until now pinned only piecewise:
    IDFT, quaternion -> matrix, Hamilton product)."""
    from helpers import reference_composition_fixture
    models, poses, want = reference_composition_fixture()
    ct_poses = [None if p is None else (p.obj_rot, p.obj_trans, p.fourier_time) for p in poses]
    xyz, scal, rot, opa, feat = ct.compose(models, ct_poses)
    for name, got in (("xyz", xyz), ("scaling", scal), ("rotation", rot), ("opacity", opa), ("features", feat)):
        np.testing.assert_allclose(got.numpy(), want[name], rtol=2e-6, atol=2e-6, err_msg=name)
    # the flip is observable: without the masks a third of the actors' Gaussians land elsewhere
    plain = [m._replace(flip=None) for m in models]
    assert np.abs(ct.compose(plain, ct_poses)[0].numpy() - want["xyz"]).max() > 1e-2
