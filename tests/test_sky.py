"""Sky cube map.

CPU part: the oracle's ray generation and face convention against reference-derived fixtures
(get_rays_torch output; cube_to_dir output), and the seamless-edge construction.
GPU part: the HIP lookup / composite / backward against the oracle.
"""
import math
import os

import numpy as np
import pytest
import torch

from gaussianrpg_amd import harness as hz
from helpers import GOLDEN
from oracle import sky_torch as st


def test_rays_match_reference_get_rays_torch():
    z = np.load(os.path.join(GOLDEN, "ref_rays.npz"))
    for n in range(2):
        H, W = (int(v) for v in z["HW%d" % n])
        K, R, T = (torch.tensor(z[k + str(n)]) for k in ("K", "R", "T"))
        ro, rd = st.get_rays(H, W, K, R, T)
        np.testing.assert_array_equal(rd.numpy(), z["rays_d%d" % n])
        np.testing.assert_array_equal(ro.numpy(), z["rays_o%d" % n])
        # perturb=True (train mode): the two torch.rand(H, W) planes re-drawn from the recorded seed
        torch.manual_seed(int(z["perturb_seed%d" % n]))
        jit = torch.stack([torch.rand(H, W), torch.rand(H, W)])
        _, rdp = st.get_rays(H, W, K, R, T, jitter=jit)
        np.testing.assert_array_equal(rdp.numpy(), z["rays_d_perturb%d" % n])
        # the product's closed form R^T K^-1 (x+.5, y+.5, 1) gives the same directions
        from gaussianrpg_amd.sky import ray_matrix
        w2c = torch.eye(4)
        w2c[:3, :3], w2c[:3, 3] = R, T
        M = ray_matrix(K, w2c).double()
        j, i = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing='ij')
        p = torch.stack([i + 0.5, j + 0.5, torch.ones_like(i)], -1) @ M.T
        p = p / p.norm(dim=-1, keepdim=True)
        assert float((p - rd.double()).abs().max()) < 2e-6


def test_face_convention_matches_reference_cube_to_dir():
    z = np.load(os.path.join(GOLDEN, "ref_cube_dir.npz"))
    x, y = z["x"].astype(np.float64), z["y"].astype(np.float64)
    for s in range(6):
        np.testing.assert_allclose(st.cube_to_dir(s, x, y), z["dirs"][s], atol=0)
        f, u, v = st.dir_to_face_uv(z["dirs"][s].astype(np.float64) * 3.7)    # scale invariant
        assert (f == s).all()
        np.testing.assert_allclose(u, x * 0.5 + 0.5, atol=1e-7)
        np.testing.assert_allclose(v, y * 0.5 + 0.5, atol=1e-7)


def test_seamless_lookup_is_continuous_across_edges():
    """A smooth function of direction stored in the cube map is reproduced to O(1/res^2) everywhere,
    including right at the edges (a lookup that clamped at face borders would show O(1/res) jumps)."""
    res = 32
    c = (np.arange(res) + 0.5) * 2.0 / res - 1.0
    gx, gy = np.meshgrid(c, c, indexing='xy')
    cube = np.zeros((6, res, res, 3))
    w = np.array([0.3, -0.5, 0.8])
    for s in range(6):
        d = st.cube_to_dir(s, gx, gy)
        d = d / np.linalg.norm(d, axis=-1, keepdims=True)
        cube[s] = 0.5 + 0.4 * (d @ w)[..., None] * np.array([1.0, 0.5, -0.7])
    rng = np.random.default_rng(0)
    dirs = rng.normal(size=(4000, 3))
    # plus directions hugging the x/z edge and a corner
    t = np.linspace(-1, 1, 200)
    dirs = np.concatenate([dirs, np.stack([np.ones_like(t), t, -1.0 + 1e-3 * t], 1),
                           np.stack([1 + 1e-3 * t, 1 - 2e-3 * t, 1.0 + 0 * t], 1)])
    got = st.texture_cube(cube.astype(np.float32), dirs)
    dn = dirs / np.linalg.norm(dirs, axis=-1, keepdims=True)
    ref = 0.5 + 0.4 * (dn @ w)[:, None] * np.array([1.0, 0.5, -0.7])
    assert np.abs(got - ref).max() < 6e-3


def _camera(Wd, Hd, yaw=0.3, pitch=-0.25):
    cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    w2c = torch.eye(4)
    w2c[:3, :3] = torch.tensor(Rx @ Ry, dtype=torch.float32)
    w2c[:3, 3] = torch.tensor([0.3, -1.2, 2.0])
    K = torch.tensor([[0.55 * Wd, 0.0, Wd / 2.0], [0.0, 0.55 * Wd, Hd / 2.0], [0.0, 0.0, 1.0]])
    return K, w2c


@pytest.mark.gpu
@pytest.mark.parametrize("res,hw,yaw,pitch", [(16, (96, 160), 0.3, -0.25), (64, (121, 203), 2.4, 0.9),
                                              (8, (64, 64), -0.78, -0.62)])
def test_sky_lookup_and_composite_match_oracle(res, hw, yaw, pitch):
    from gaussianrpg_amd.sky import SkyCubeMap
    dev = torch.device("cuda:0")
    Hd, Wd = hw
    K, w2c = _camera(Wd, Hd, yaw, pitch)
    g = torch.Generator().manual_seed(res)
    sky = SkyCubeMap(res, white_background=(res == 8)).to(dev)
    cube = torch.rand(6, res, res, 3, generator=g) * 1.3 - 0.15      # some texels outside [0,1]: clamp matters
    with torch.no_grad():
        sky.sky_cube_map.copy_(cube.to(dev))
    acc = torch.rand(1, Hd, Wd, generator=g)
    acc[:, : Hd // 3] = 1.0 - 5e-4 * torch.rand(1, Hd // 3, Wd, generator=g)   # below the 1e-3 mask threshold
    rgb = torch.rand(3, Hd, Wd, generator=g)
    fill = 1.0 if res == 8 else 0.0
    ref_sky = st.sky_color(cube.numpy(), K, w2c, Hd, Wd, acc=acc.numpy(), fill=fill)
    got_sky = sky(K, w2c, Hd, Wd, acc.to(dev)).detach().cpu().numpy()
    assert np.abs(got_sky - ref_sky).max() < 2e-5
    ref_all = st.sky_color(cube.numpy(), K, w2c, Hd, Wd, acc=None)
    got_all = sky(K, w2c, Hd, Wd, None).detach().cpu().numpy()
    assert np.abs(got_all - ref_all).max() < 2e-5
    for train in (False, True):
        got = sky.composite(rgb.to(dev), acc.to(dev), K, w2c, train=train).detach().cpu().numpy()
        ref = st.composite(rgb.numpy(), acc.numpy(), ref_sky, clamp=not train)
        assert np.abs(got - ref).max() < 3e-5
    # the views really look across several faces
    _, rays = st.get_rays(Hd, Wd, K, w2c[:3, :3], w2c[:3, 3])
    faces = np.unique(st.dir_to_face_uv(rays.double().numpy().reshape(-1, 3))[0])
    assert len(faces) >= 2


@pytest.mark.gpu
def test_sky_backward():
    """The composite is linear in the cube map (no texel clamps here), so <grad_cube, D> must equal
    the change of the loss along D; grad_acc = -sum_c sky_c g_c; grad_rgb = g."""
    from gaussianrpg_amd.sky import SkyCubeMap
    dev = torch.device("cuda:0")
    res, Hd, Wd = 16, 72, 120
    K, w2c = _camera(Wd, Hd, 1.1, 0.4)
    g = torch.Generator().manual_seed(4)
    sky = SkyCubeMap(res).to(dev)
    with torch.no_grad():
        sky.sky_cube_map.copy_((0.2 + 0.6 * torch.rand(6, res, res, 3, generator=g)).to(dev))
    acc = (0.9 * torch.rand(1, Hd, Wd, generator=g)).to(dev).requires_grad_(True)
    rgb = torch.rand(3, Hd, Wd, generator=g).to(dev).requires_grad_(True)
    wgt = torch.randn(3, Hd, Wd, generator=g).to(dev)
    out = sky.composite(rgb, acc, K, w2c, train=True)
    (out * wgt).sum().backward()
    torch.cuda.synchronize()
    assert torch.allclose(rgb.grad, wgt)
    sky_plane = torch.tensor(st.sky_color(sky.sky_cube_map.detach().cpu().numpy(), K, w2c, Hd, Wd, acc=None),
                             dtype=torch.float32).to(dev)
    ref_gacc = -(sky_plane * wgt).sum(0, keepdim=True)
    assert float((acc.grad - ref_gacc).abs().max()) < 2e-4
    D = torch.randn(6, res, res, 3, generator=g).to(dev)
    lhs = float((sky.sky_cube_map.grad.double() * D.double()).sum())
    with torch.no_grad():
        base = float((sky.composite(rgb, acc, K, w2c, train=True).double() * wgt.double()).sum())
        sky.sky_cube_map.add_(1e-2 * D)
        moved = float((sky.composite(rgb, acc, K, w2c, train=True).double() * wgt.double()).sum())
    rhs = (moved - base) / 1e-2
    assert abs(lhs - rhs) <= 2e-3 * max(1.0, abs(rhs)), (lhs, rhs)
    # the standalone lookup is differentiable w.r.t. the cube map too
    sky.sky_cube_map.grad = None
    s2 = sky(K, w2c, Hd, Wd, None)
    (s2 * wgt).sum().backward()
    assert float(sky.sky_cube_map.grad.abs().sum()) > 0


class _RefLikeCamera:
    """The attributes SkyCubeMap.forward reads off a reference Camera (sky_cubemap.py:80-89)."""

    def __init__(self, K, w2c, H, W, sky_mask=None):
        self.K, self.world_view_transform = K, w2c.transpose(0, 1).contiguous()
        self.image_height, self.image_width = H, W
        if sky_mask is not None:
            self.original_sky_mask = sky_mask


@pytest.mark.gpu
def test_sky_train_mode_mask_and_jitter_match_oracle():
    """Train branch of SkyCubeMap.forward (sky_cubemap.py:80-82,91-92): the fetch mask is the camera's
    sky mask with the first 50 rows set -- NOT the acc rule -- and rays are jittered inside their
    pixels; called as the reference's renderer does, ``sky(camera, acc)``, with device-resident K and
    world_view_transform (no host read of the camera)."""
    from gaussianrpg_amd.sky import SkyCubeMap, train_sky_mask
    dev = torch.device("cuda:0")
    res, Hd, Wd = 32, 120, 176
    K, w2c = _camera(Wd, Hd, 0.9, -0.4)
    g = torch.Generator().manual_seed(11)
    sky = SkyCubeMap(res, mode="train").to(dev)
    cube = torch.rand(6, res, res, 3, generator=g) * 1.2 - 0.1
    with torch.no_grad():
        sky.sky_cube_map.copy_(cube.to(dev))
    acc = torch.rand(1, Hd, Wd, generator=g)
    rgb = torch.rand(3, Hd, Wd, generator=g)
    sky_mask = torch.rand(1, Hd, Wd, generator=g) > 0.6
    jitter = torch.rand(2, Hd, Wd, generator=g)
    cam = _RefLikeCamera(K.to(dev), w2c.to(dev), Hd, Wd, sky_mask.to(dev))
    m = train_sky_mask(sky_mask)
    assert m[:50].all() and not m[50:].all() and sky_mask[0, :50].float().mean() < 0.9
    ref_sky = st.sky_color(cube.numpy(), K, w2c, Hd, Wd, acc=acc.numpy(), mask=m.numpy(), jitter=jitter)
    got = sky(cam, acc.to(dev), jitter=jitter.to(dev)).detach().cpu().numpy()
    assert np.abs(got - ref_sky).max() < 2e-5
    # the acc rule would have given something else: the mask really is the camera's
    other = st.sky_color(cube.numpy(), K, w2c, Hd, Wd, acc=acc.numpy(), jitter=jitter)
    assert np.abs(other - ref_sky).max() > 0.1
    assert not cam.original_sky_mask[0, :50].all()       # the camera's own tensor is left alone
    # composite in train mode: unclamped, same mask / jitter, differentiable
    out = sky.composite(rgb.to(dev), acc.to(dev), camera=cam, jitter=jitter.to(dev))
    ref = st.composite(rgb.numpy(), acc.numpy(), ref_sky, clamp=False)
    assert np.abs(out.detach().cpu().numpy() - ref).max() < 3e-5
    out.sum().backward()
    gcube = sky.sky_cube_map.grad
    assert float(gcube.abs().sum()) > 0
    # gradient mass = sum over fetching, unclamped pixels of (1 - acc): linearity in the cube map
    sky.sky_cube_map.grad = None
    # default jitter: drawn per call, inside the pixel -> two calls differ, both near the centre render
    a = sky(cam, acc.to(dev)).detach()
    b = sky(cam, acc.to(dev)).detach()
    assert not torch.equal(a, b)
    # a camera without a sky mask falls back to the acc rule, still jittered (sky_cubemap.py:83-85)
    cam2 = _RefLikeCamera(K.to(dev), w2c.to(dev), Hd, Wd)
    got2 = sky(cam2, acc.to(dev), jitter=jitter.to(dev)).detach().cpu().numpy()
    assert np.abs(got2 - other).max() < 2e-5
    # evaluation mode through the camera form == the explicit form
    sky.mode = "evaluate"
    e1 = sky(cam, acc.to(dev)).detach()
    e2 = sky(K, w2c, Hd, Wd, acc.to(dev)).detach()
    assert float((e1 - e2).abs().max()) < 1e-6
