"""Sky cube map without nvdiffrast (additive API).

The reference's ``SkyCubeMap`` (lib/models/sky_cubemap.py:14-122) needs ``nvdiffrast.torch`` -- a
CUDA-only, un-vendored dependency -- for one call, ``dr.texture(cube, rays, filter_mode='linear',
boundary_mode='cube')``; every example config sets ``include_sky: true``, so on a ROCm machine
``lib.models`` does not even import.  This module provides the same object on the HIP library:

    sky = SkyCubeMap(resolution=1024, white_background=False).cuda()
    sky_color = sky(K, w2c, H, W, acc)                   # [3,H,W], differentiable w.r.t. the cube map and acc
    rgb = sky.composite(rgb, acc, K, w2c, train=False)   # rgb + sky * (1 - acc) (+ clamp), ONE launch

``forward`` mirrors ``SkyCubeMap.forward`` (eval branch of the mask: ``(1 - acc) > 1e-3``;
``get_rays_torch`` without perturbation); ``composite`` fuses it with
``StreetGaussianRenderer.render``'s ``rgb + sky_color * (1 - acc)`` and the eval-mode clamp
(lib/models/street_gaussian_renderer.py:106-116).  No fixed 1080x1920 scratch image
(sky_cubemap.py:29-34): any frame size works.
"""
import torch
import torch.nn as nn

from .rasterizer import _C


def ray_matrix(K: torch.Tensor, w2c: torch.Tensor) -> torch.Tensor:
    """[3,3] CPU float32 matrix M with ray(x, y) ~ M (x + 0.5, y + 0.5, 1): get_rays_torch
    (lib/utils/graphics_utils.py:186-207) computes normalize(((K^-1 p) - T) R - rays_o) with
    rays_o = -R^T T, which is R^T K^-1 p before the normalisation.  Evaluated in float64, rounded once."""
    K64 = K.detach().double().cpu()
    R64 = w2c.detach().double().cpu()[:3, :3]
    return (R64.transpose(0, 1) @ torch.linalg.inv(K64)).float().contiguous()


class _SkyLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cube, acc, rm, fill, H, W):
        _, sky = _C.sky_composite(cube, rm, float(fill), False, None, acc, int(H), int(W), True)
        ctx.save_for_backward(cube, acc if acc is not None else torch.Tensor([]))
        ctx.rm, ctx.fill, ctx.has_acc = rm, float(fill), acc is not None
        return sky

    @staticmethod
    def backward(ctx, grad_sky):
        # sky = clamp(fetch) where the mask is on: the lookup's own gradient is the composite's with
        # (1 - acc) replaced by 1, i.e. sky_backward with acc = None restricted to the mask.  The
        # kernel applies the mask itself when acc is given, and scales by (1 - acc): divide it out by
        # handing it a pre-scaled gradient.
        cube, acc = ctx.saved_tensors
        if ctx.has_acc:
            tr = (1.0 - acc).clamp_min(1e-12)
            g = grad_sky / tr
            grad_cube, _ = _C.sky_backward(cube, ctx.rm, ctx.fill, acc, g.contiguous())
        else:
            grad_cube, _ = _C.sky_backward(cube, ctx.rm, ctx.fill, None, grad_sky.contiguous())
        return grad_cube, None, None, None, None, None


class _SkyComposite(torch.autograd.Function):
    """rgb + clamp(sky, 0, 1) * (1 - acc), unclamped output (train mode)."""

    @staticmethod
    def forward(ctx, cube, rgb, acc, rm, fill):
        out, _ = _C.sky_composite(cube, rm, float(fill), False, rgb, acc, rgb.shape[1], rgb.shape[2], False)
        ctx.save_for_backward(cube, acc)
        ctx.rm, ctx.fill = rm, float(fill)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        cube, acc = ctx.saved_tensors
        grad_cube, grad_acc = _C.sky_backward(cube, ctx.rm, ctx.fill, acc, grad_out.contiguous())
        return grad_cube, grad_out, grad_acc.reshape(acc.shape), None, None


class SkyCubeMap(nn.Module):
    """Drop-in for lib/models/sky_cubemap.py:SkyCubeMap (same parameter name and initial value)."""

    def __init__(self, resolution: int = 1024, white_background: bool = False):
        super().__init__()
        self.resolution = int(resolution)
        self.white_background = bool(white_background)
        eps = 1e-3
        if white_background:      # sky_cubemap.py:22-25
            base = torch.ones(6, self.resolution, self.resolution, 3) * (1.0 - eps)
        else:
            base = torch.zeros(6, self.resolution, self.resolution, 3) + eps
        self.sky_cube_map = nn.Parameter(base.float())

    @property
    def fill(self) -> float:
        return 1.0 if self.white_background else 0.0

    def forward(self, K, w2c, image_height, image_width, acc=None):
        """[3,H,W] sky colour: the cube map seen along every pixel's ray where (1 - acc) > 1e-3, the
        background fill elsewhere, clamped to [0,1] (sky_cubemap.py:77-122, evaluation mask)."""
        rm = ray_matrix(K, w2c)
        a = None if acc is None else acc.detach()
        return _SkyLookup.apply(self.sky_cube_map, a, rm, self.fill, image_height, image_width)

    def composite(self, rgb, acc, K, w2c, train: bool = False):
        """``rgb + sky * (1 - acc)`` (street_gaussian_renderer.py:106-110), clamped to [0,1] outside
        train mode (:115-116), fused into one launch.  In train mode the result is differentiable
        w.r.t. the cube map, rgb and acc."""
        rm = ray_matrix(K, w2c)
        if train:
            return _SkyComposite.apply(self.sky_cube_map, rgb, acc, rm, self.fill)
        with torch.no_grad():
            out, _ = _C.sky_composite(self.sky_cube_map, rm, self.fill, True, rgb, acc, rgb.shape[1],
                                      rgb.shape[2], False)
        return out
