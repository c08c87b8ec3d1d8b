"""Fused scene-graph composition.

CPU part (-m "not gpu"): the IDFT weights against the reference-derived fixture.
GPU part (-m gpu): (1) grpg_compose == the torch restatement of the reference's getters within
float tolerance; (2) the fused ComposedRasterizer == the classic GaussianRasterizer fed with
grpg_compose's own output, BIT FOR BIT on every output (the composition feeds a discontinuous
function -- ceil of the 3-sigma radius, tile rectangles -- so integer parity is only meaningful
against the very same activated values).
"""
import math
import os

import numpy as np
import pytest
import torch

from gaussianrpg_amd import harness as hz
from helpers import GOLDEN


def test_idft_weights_match_reference_fixture():
    from gaussianrpg_amd.composed import idft_weights
    from oracle import compose_torch as ct
    z = np.load(os.path.join(GOLDEN, "ref_idft.npz"))
    for d in (1, 2, 3, 5, 8):
        for i, t in enumerate(z["times"]):
            ref = z["dim%d" % d][i]
            np.testing.assert_array_equal(np.array(idft_weights(float(t), d), np.float32), ref)
            np.testing.assert_array_equal(ct.idft(float(t), d)[0].numpy(), ref)


def test_batched_idft_rows_match_reference_fixture():
    """The per-frame form (_idft_rows: all models' weights from one column of times) against the same fixture: the same
    bits as one reference call per model, zero beyond a model's own dimension."""
    from gaussianrpg_amd.composed import MAX_FOURIER, _idft_rows
    z = np.load(os.path.join(GOLDEN, "ref_idft.npz"))
    dims = [1, 2, 3, 5, 8]
    for i, t in enumerate(z["times"]):
        for order in (dims, dims[::-1], [5, 5, 1, 8]):
            rows = _idft_rows([float(t)] * len(order), order).numpy()
            assert rows.shape == (len(order), MAX_FOURIER) and rows.dtype == np.float32
            for r, d in zip(rows, order):
                np.testing.assert_array_equal(r[:d].view(np.uint32), z["dim%d" % d][i].astype(np.float32).view(np.uint32))
                assert not r[d:].view(np.uint32).any()      # +0.0, not -0.0
    # different times per row
    ts = [float(x) for x in z["times"][:5]]
    rows = _idft_rows(ts, dims).numpy()
    for k, (t_i, d) in enumerate(zip(range(5), dims)):
        np.testing.assert_array_equal(rows[k][:d], z["dim%d" % d][t_i])


def _to(m, dev, grad=False):
    """ModelParams on a device (the optional flip mask may be None); grad: fresh leaves."""
    f = lambda t: t.to(dev).clone().requires_grad_(True) if grad else t.to(dev)   # noqa: E731
    return type(m)(*(f(t) for t in m[:6]), None if m.flip is None else m.flip.to(dev))


def _logit(p):
    return torch.log(p / (1 - p))


def _scene_graph(sh_degree=1, nb=60_000, actors=((4000, 5), (2500, 3), (6000, 1)), seed=5):
    """Background = a street scene expressed in RAW parameters; actors = boxes of Gaussians in
    object-local coordinates with tracked poses."""
    from gaussianrpg_amd.composed import ActorPose, ModelParams
    g = torch.Generator().manual_seed(seed)
    sc = hz.street_scene(nb, seed=seed, sh_degree=sh_degree)
    M = (sh_degree + 1) ** 2
    models = [ModelParams(sc.means3D, torch.log(sc.scales), sc.rotations * (0.5 + torch.rand(nb, 1, generator=g)),
                          _logit(sc.opacity.clamp(1e-4, 1 - 1e-4)), sc.shs[:, :1].contiguous(),
                          sc.shs[:, 1:].contiguous())]
    poses = [None]
    for k, (n, F) in enumerate(actors):
        xyz = (torch.rand(n, 3, generator=g) - 0.5) * torch.tensor([4.5, 1.6, 2.0])
        scaling = math.log(0.05) + 0.5 * torch.randn(n, 3, generator=g)
        rotation = torch.randn(n, 4, generator=g)
        opacity = 1.0 + 2.0 * torch.randn(n, 1, generator=g)
        fdc = 0.5 * torch.randn(n, F, 3, generator=g)
        frest = 0.15 * torch.randn(n, M - 1, 3, generator=g)
        # the first actor carries this iteration's flip mask (training symmetry prior, flip_prob 0.5)
        flip = (torch.rand(n, generator=g) < 0.5) if k == 0 else None
        models.append(ModelParams(xyz, scaling, rotation, opacity, fdc, frest, flip))
        q = torch.randn(4, generator=g)
        q = q / q.norm() * (1.0 + 0.01 * k)        # obj_rot is not exactly unit in practice
        trans = torch.tensor([-6.0 + 5.0 * k, 0.8, 12.0 + 9.0 * k])
        poses.append(ActorPose(q.tolist(), trans.tolist(), 0.2 + 0.3 * k))
    return models, poses


@pytest.mark.gpu
@pytest.mark.parametrize("sh_degree", [1, 2, 0])
def test_compose_matches_torch_restatement(sh_degree):
    from gaussianrpg_amd.composed import compose
    from oracle import compose_torch as ct
    dev = torch.device("cuda:0")
    models, poses = _scene_graph(sh_degree)
    got = compose([_to(m, dev) for m in models], poses)
    ref = ct.compose(models, [None if p is None else (p.obj_rot, p.obj_trans, p.fourier_time) for p in poses])
    names = ("means3D", "scales", "rotations", "opacity", "shs")
    P = sum(m.xyz.shape[0] for m in models)
    for n, a, b in zip(names, got, ref):
        a = a.cpu()
        assert a.shape == b.shape and a.shape[0] == P, (n, a.shape, b.shape)
        torch.testing.assert_close(a, b, rtol=2e-6, atol=2e-6, msg=lambda m, n=n: n + ": " + m)
    # the background passes through exactly where no arithmetic is involved
    nb = models[0].xyz.shape[0]
    assert torch.equal(got[0][:nb].cpu(), models[0].xyz)
    assert torch.equal(got[4][:nb, 1:].cpu(), models[0].features_rest)


@pytest.mark.gpu
@pytest.mark.parametrize("sh_degree", [1, 2])
def test_fused_forward_equals_classic_on_composed_tensors(sh_degree):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from gaussianrpg_amd.composed import ComposedRasterizer, compose
    dev = torch.device("cuda:0")
    models, poses = _scene_graph(sh_degree)
    models = [_to(m, dev) for m in models]
    cam = hz.trajectory_camera(2, W=960, H=640, device=dev)
    bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
    rs = GaussianRasterizationSettings(**hz.settings_kwargs(cam, sh_degree, bg=bg))
    fused = ComposedRasterizer(rs)
    c1, r1, d1, a1 = fused(models, poses)
    means, scales, rots, opac, shs = compose(models, poses)
    c2, r2, d2, a2, _ = GaussianRasterizer(rs)(means3D=means, means2D=None, opacities=opac, shs=shs,
                                               scales=scales, rotations=rots)
    torch.cuda.synchronize()
    assert int((r1 > 0).sum()) > 10_000
    nb = models[0].xyz.shape[0]
    assert int((r1[nb:] > 0).sum()) > 1000, "the actors must be in view for this test to mean anything"
    assert torch.equal(r1, r2)
    assert torch.equal(c1, c2) and torch.equal(d1, d2) and torch.equal(a1, a2)
    # a second frame with other poses through the same module (capacity hint path)
    from gaussianrpg_amd.composed import ActorPose
    poses2 = [None] + [ActorPose(p.obj_rot, [p.obj_trans[0] + 0.5, p.obj_trans[1], p.obj_trans[2] - 1.0],
                                 p.fourier_time + 0.1) for p in poses[1:]]
    c3, r3, d3, a3 = fused(models, poses2)
    m3 = compose(models, poses2)
    c4, r4, d4, a4, _ = GaussianRasterizer(rs)(means3D=m3[0], means2D=None, opacities=m3[3], shs=m3[4],
                                               scales=m3[1], rotations=m3[2])
    assert torch.equal(c3, c4) and torch.equal(r3, r4) and not torch.equal(c1, c3)


@pytest.mark.gpu
@pytest.mark.parametrize("sh_degree", [1, 2])
def test_composed_layers_equal_flat_layers_and_three_calls(sh_degree):
    """grpg_forward_composed_layers: scene-graph composition + the three renders of render_all from the raw
    parameters.  Bit-identical to the layered forward on compose()'s own output (class = "posed model"), which
    tests/test_gpu_layers.py ties to the three separate op calls; an explicit model selection likewise."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from gaussianrpg_amd.composed import ComposedRasterizer, compose
    dev = torch.device("cuda:0")
    models, poses = _scene_graph(sh_degree)
    models = [_to(m, dev) for m in models]
    cam = hz.trajectory_camera(2, W=960, H=640, device=dev)
    rs = GaussianRasterizationSettings(**hz.settings_kwargs(cam, sh_degree, bg=torch.tensor([0.2, 0.1, 0.3], device=dev)))
    fused = ComposedRasterizer(rs)
    means, scales, rots, opac, shs = compose(models, poses)
    counts = [m.xyz.shape[0] for m in models]
    keys = ("color", "depth", "alpha", "radii", "color_background", "alpha_background", "color_object", "alpha_object")
    for flags in (None, [False, True, False, True]):
        one = fused.forward_layers(models, poses, object_models=flags)
        per_model = [p is not None for p in poses] if flags is None else flags
        cls = torch.cat([torch.full((n,), bool(f), dtype=torch.bool, device=dev) for n, f in zip(counts, per_model)])
        flat = GaussianRasterizer(rs).forward_layers(means, opac, cls, shs=shs, scales=scales, rotations=rots)
        torch.cuda.synchronize()
        for k in keys:
            assert torch.equal(one[k], flat[k]), (k, flags)
        assert float(one["alpha_object"].max()) > 0.5 and float((one["color"] - one["color_background"]).abs().max()) > 0.05
    # the composition plane is the plain fused forward's
    c1, r1, d1, a1 = fused(models, poses)
    assert torch.equal(c1, one["color"]) and torch.equal(d1, one["depth"]) and torch.equal(a1, one["alpha"])
    # argument check
    with pytest.raises(RuntimeError, match="object_model"):
        fused.forward_layers(models, poses, object_models=[True])


@pytest.mark.gpu
def test_single_gaussian_actors():
    """Actors of ONE Gaussian each behind a small one (a workgroup of the preprocess kernels then spans
    four models; empty models are rejected by the binding, test_composed_argument_errors): fused ==
    classic bit for bit, and the training backward returns gradients of every tensor's own shape."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from gaussianrpg_amd.composed import ComposedRasterizer, compose
    dev = torch.device("cuda:0")
    models, poses = _scene_graph(1, nb=5000, actors=((300, 5), (1, 3), (1, 2)))
    models = [_to(m, dev, grad=True) for m in models]
    cam = hz.trajectory_camera(2, W=320, H=208, device=dev)
    rs = GaussianRasterizationSettings(**hz.settings_kwargs(cam, 1))
    fused = ComposedRasterizer(rs)
    with torch.no_grad():
        c1, r1, d1, a1 = fused(models, poses)
        means, scales, rots, opac, shs = compose(models, poses)
        c2, r2, d2, a2, _ = GaussianRasterizer(rs)(means3D=means, means2D=None, opacities=opac, shs=shs,
                                                   scales=scales, rotations=rots)
    assert torch.equal(r1, r2) and torch.equal(c1, c2) and torch.equal(d1, d2) and torch.equal(a1, a2)
    c, r, d, a = fused(models, poses)
    (c.sum() + d.sum() + a.sum()).backward()
    torch.cuda.synchronize()
    for m in models:
        for t in m[:6]:
            assert t.grad is not None and t.grad.shape == t.shape and bool(torch.isfinite(t.grad).all())


@pytest.mark.gpu
def test_composed_argument_errors():
    from gaussianrpg_amd.composed import ComposedRasterizer
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    dev = torch.device("cuda:0")
    models, poses = _scene_graph(1, nb=2000, actors=((300, 2),))
    models = [_to(m, dev) for m in models]
    cam = hz.trajectory_camera(0, W=128, H=96, device=dev)
    fused = ComposedRasterizer(GaussianRasterizationSettings(**hz.settings_kwargs(cam, 1)))
    with pytest.raises(ValueError):
        fused(models, poses[:1])
    bad = [models[0], models[1]._replace(features_rest=models[1].features_rest[:, :1])]
    with pytest.raises(RuntimeError, match="same number of SH coefficients"):
        fused(bad, poses)


@pytest.mark.gpu
def test_compose_matches_reference_quaternion_fixture():
    """grpg_compose against values produced by the reference's own quaternion helpers
    (tests/golden/ref_quat.npz: quaternion_to_matrix_numpy R/lib/utils/general_utils.py:103-122,
    quaternion_raw_multiply :220-238): actor means = R(obj_rot) x + t
    (street_gaussian_model.py:352-356) and rotations = normalize(obj_rot (x) normalize(q)) (:330-336)."""
    from gaussianrpg_amd.composed import ActorPose, ModelParams, compose
    z = np.load(os.path.join(GOLDEN, "ref_quat.npz"))
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    n = 500
    xyz = torch.randn(n, 3, generator=g)
    for k in range(8):
        obj_rot, Rref = z["q"][k], z["R"][k]                 # R of the normalised quaternion, float64
        a, b = z["a"][k], torch.tensor(z["b"][k])            # one fixture row: a (x) b = ab[k]
        rot_local = b.reshape(1, 4).repeat(n, 1).contiguous()
        m = ModelParams(xyz, torch.zeros(n, 3), rot_local, torch.zeros(n, 1), torch.zeros(n, 1, 3),
                        torch.zeros(n, 3, 3))
        trans = [0.5 * k, -1.0, 3.0]
        for rot, check in ((obj_rot, "matrix"), (a, "product")):
            got = compose([_to(m, dev)],
                          [ActorPose([float(v) for v in rot], trans, 0.0)])
            if check == "matrix":
                ref = xyz.double().numpy() @ Rref.T + np.array(trans)
                np.testing.assert_allclose(got[0].cpu().numpy(), ref, rtol=0, atol=5e-6)
            else:
                # the product is bilinear: a (x) (b/|b|) = (a (x) b)/|b|, and the result is normalised
                ref = z["ab"][k].astype(np.float64)
                ref = ref / np.linalg.norm(ref)
                np.testing.assert_allclose(got[2].cpu().numpy(), np.tile(ref, (n, 1)), rtol=0, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("sh_degree", [1, 2])
def test_fused_training_gradients_match_autograd_through_the_restatement(sh_degree):
    """Training through the fused composition (C ABI grpg_forward_composed_flags + grpg_backward_composed):
    gradients with respect to every model's RAW parameters, the actors' poses and means2D against
    torch.autograd through oracle/compose_torch.py (the restatement of the reference's getters,
    street_gaussian_model.py:296-453) followed by the classic op -- whose own backward is
    oracle-checked in tests/test_gpu_backward.py."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from gaussianrpg_amd.composed import ActorPose, ComposedRasterizer
    from oracle import compose_torch as ct
    dev = torch.device("cuda:0")
    models, poses = _scene_graph(sh_degree, nb=30_000, actors=((3000, 5), (2000, 1)), seed=9)
    # the symmetry prior of the shipped training configs (flip_prob: 0.5): half of the first actor's
    # Gaussians are flipped this iteration, the second actor has no mask
    gf = torch.Generator().manual_seed(17)
    models[1] = models[1]._replace(flip=torch.rand(models[1].xyz.shape[0], generator=gf) < 0.5)
    cam = hz.trajectory_camera(2, W=480, H=320, device=dev)
    bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
    rs = GaussianRasterizationSettings(**hz.settings_kwargs(cam, sh_degree, bg=bg))
    P = sum(m.xyz.shape[0] for m in models)
    g = torch.Generator().manual_seed(3)
    gc = torch.randn(3, 320, 480, generator=g).to(dev)
    gd = (0.1 * torch.randn(1, 320, 480, generator=g)).to(dev)
    ga = torch.randn(1, 320, 480, generator=g).to(dev)

    def leaves():
        ms = [_to(m, dev, grad=True) for m in models]
        ps = [None if p is None else ActorPose(torch.tensor(p.obj_rot, device=dev, requires_grad=True),
                                               torch.tensor(p.obj_trans, device=dev, requires_grad=True),
                                               p.fourier_time) for p in poses]
        m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
        return ms, ps, m2d

    def loss_of(color, depth, alpha):
        return (color * gc).sum() + (depth * gd).sum() + (alpha * ga).sum()

    # fused
    ms, ps, m2d = leaves()
    color, radii, depth, alpha = ComposedRasterizer(rs)(ms, ps, means2D=m2d)
    assert color.requires_grad and int((radii > 0).sum()) > 5000
    nb = models[0].xyz.shape[0]
    assert int((radii[nb:] > 0).sum()) > 500, "the actors must be in view"
    loss_of(color, depth, alpha).backward()
    # restatement + classic op
    mr, pr, m2r = leaves()
    means, scales, rots, opac, shs = ct.compose(
        mr, [None if p is None else (p.obj_rot, p.obj_trans, p.fourier_time) for p in pr])
    c2, r2, d2, a2, _ = GaussianRasterizer(rs)(means3D=means, means2D=m2r, opacities=opac, shs=shs,
                                               scales=scales, rotations=rots)
    loss_of(c2, d2, a2).backward()
    torch.cuda.synchronize()

    def close(name, got, ref, tol=2e-3):
        got, ref = got.double(), ref.double()
        den = float(ref.norm()) + 1e-30
        rel = float((got - ref).norm()) / den
        assert rel <= tol, "%s: relative L2 %.3e" % (name, rel)
        assert float(ref.abs().max()) > 0, name + ": reference gradient is all zero"

    close("means2D", m2d.grad, m2r.grad)
    for i, (a, b) in enumerate(zip(ms, mr)):
        for f in a._fields[:6]:
            ga_, gb_ = getattr(a, f).grad, getattr(b, f).grad
            if gb_ is None or float(gb_.abs().max()) == 0.0:    # e.g. SH bands above the active degree
                assert ga_ is None or float(ga_.abs().max()) == 0.0, (i, f)
                continue
            close("model %d %s" % (i, f), ga_, gb_)
    for i, (a, b) in enumerate(zip(ps, pr)):
        if a is None:
            continue
        close("pose %d rot" % i, a.obj_rot.grad, b.obj_rot.grad, 5e-3)
        close("pose %d trans" % i, a.obj_trans.grad, b.obj_trans.grad, 5e-3)
    # culled Gaussians get exactly zero gradient in every raw parameter
    culled = (radii == 0)[:nb]
    assert bool(culled.any()) and float(ms[0].xyz.grad[culled].abs().max()) == 0.0
    assert float(ms[0].scaling.grad[culled].abs().max()) == 0.0
    # evaluation path unchanged: no graph, same image
    with torch.no_grad():
        c3, r3, d3, a3 = ComposedRasterizer(rs)(ms, ps)
    assert not c3.requires_grad and torch.equal(c3, color.detach()) and torch.equal(r3, radii)


@pytest.mark.gpu
def test_compose_matches_the_references_own_getters():
    """grpg_compose (the HIP code the fused rasterizer runs) against tests/golden/ref_compose.npz: what
    StreetGaussianModel.get_xyz / get_rotation / get_scaling / get_opacity / get_features returned IN the
    reference's code for a background model and two posed, partly flipped actors
    (street_gaussian_model.py:296-453, gaussian_model_actor.py:73-82; make_golden.py part_a_compose)."""
    from gaussianrpg_amd.composed import compose
    from helpers import reference_composition_fixture
    models, poses, want = reference_composition_fixture()
    dev = torch.device("cuda:0")
    got = compose([_to(m, dev) for m in models], poses)
    for name, t in zip(("xyz", "scaling", "rotation", "opacity", "features"), got):
        np.testing.assert_allclose(t.cpu().numpy(), want[name], rtol=3e-6, atol=3e-6, err_msg=name)
