"""Fused scene-graph composition in front of the rasterizer (additive API).

Every frame the reference's ``StreetGaussianModel`` rebuilds the op's flat inputs in PyTorch
(lib/models/street_gaussian_model.py:296-453): per model ``exp`` / ``sigmoid`` / ``normalize`` of the
raw parameters (gaussian_model.py:224-251), per visible actor a rigid transform of the means and a
quaternion product for the rotations with the tracked pose (street_gaussian_model.py:318-365), an
inverse DFT of the actor's Fourier colour coefficients (gaussian_model_actor.py:73-82) and a
``torch.cat`` over the models -- about twenty small kernels and two extra passes over the op's whole
input.  ``ComposedRasterizer`` hands the models' RAW parameter tensors and the per-frame poses to
``_C.rasterize_gaussians_composed`` instead; the arithmetic happens inside the HIP preprocess and the
concatenated tensors are never materialised.

    models = [ModelParams(...background...), ModelParams(...actor 1...), ...]
    poses  = [None, ActorPose(obj_rot, obj_trans, fourier_time), ...]
    color, radii, depth, alpha = ComposedRasterizer(raster_settings)(models, poses)

Evaluation (``torch.no_grad()`` or nothing requires grad): forward only.  TRAINING (round 3): when
any raw parameter tensor -- or an actor's ``obj_rot`` / ``obj_trans`` given as a tensor -- requires
grad, the call goes through an autograd function whose backward (C ABI ``grpg_backward_composed``)
returns the gradients with respect to the RAW parameters (``_xyz``, ``_scaling``, ``_rotation``,
``_opacity``, ``_features_dc``, ``_features_rest``) and the poses: the chain rule through exp /
sigmoid / normalize, the Fourier DC sum, the rigid transform and the quaternion product runs inside
the preprocess backward kernel.  ``means2D`` (zeros ``[P,3]`` requiring grad, as
street_gaussian_renderer.py:157-162 creates it) receives the densification statistic.  The flip
augmentation of rigid actors (street_gaussian_model.py:286-293) is ``ModelParams.flip``.  Not fused
(stay in the caller's PyTorch): the semantic concatenation (:420-435) and pose-correction modules
-- a corrected pose is simply an ``obj_rot`` / ``obj_trans`` tensor with a graph behind it.
"""
import math
from typing import List, NamedTuple, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from .rasterizer import GaussianRasterizationSettings, _C

MAX_FOURIER = 8     # GRPG_MAX_FOURIER


class ModelParams(NamedTuple):
    """Raw (pre-activation) parameters of one Gaussian model, as the reference stores them
    (gaussian_model.py:36-47): ``_xyz [N,3]``, ``_scaling [N,3]`` (log), ``_rotation [N,4]``,
    ``_opacity [N,1]`` (logit), ``_features_dc [N,F,3]`` (F = fourier_dim, 1 for the background),
    ``_features_rest [N,M-1,3]``.  ``flip`` (optional, bool ``[N]``): this iteration's flip mask of a
    rigid actor -- the training symmetry prior, ``torch.rand_like(xyz[:, 0]) < flip_prob``
    (street_gaussian_model.py:286-293); flipped Gaussians have their local y mirrored and their local
    rotation pre-multiplied by the quaternion of diag(-1, 1, -1) (:57-61, :332, :360)."""
    xyz: torch.Tensor
    scaling: torch.Tensor
    rotation: torch.Tensor
    opacity: torch.Tensor
    features_dc: torch.Tensor
    features_rest: torch.Tensor
    flip: Optional[torch.Tensor] = None


class ActorPose(NamedTuple):
    """Pose of one actor for one frame: ``obj_rot`` (w,x,y,z) and ``obj_trans`` in world space, ego
    pose already applied (street_gaussian_model.py:268-273); ``fourier_time`` =
    fourier_scale * (frame - start_frame) / (end_frame - start_frame) (gaussian_model_actor.py:74-75).
    ``obj_rot`` / ``obj_trans`` are sequences of floats or tensors ([4] / [3]); tensors that require
    grad (optimised tracking, street_gaussian_model.py:319-320) receive gradients in training."""
    obj_rot: Sequence[float]
    obj_trans: Sequence[float]
    fourier_time: float = 0.0


def idft_weights(time: float, dim: int) -> List[float]:
    """IDFT(time, dim) of lib/utils/sh_utils.py:120-130: even k -> cos(pi t k), odd k -> sin(pi t (k+1)),
    evaluated in float32 like the reference (pinned by tests/golden/ref_idft.npz)."""
    t = torch.tensor(float(time)).view(-1, 1).float()
    idft = torch.zeros(1, dim)
    indices = torch.arange(dim)
    even, odd = indices[::2], indices[1::2]
    idft[:, even] = torch.cos(torch.pi * t * even)
    idft[:, odd] = torch.sin(torch.pi * t * (odd + 1))
    return [float(v) for v in idft[0]]


_ZERO_ROW = [0.0] * 8
_NO_FLIP = torch.empty(0, dtype=torch.bool)
_IDFT_CONST = {}   # dims tuple -> (multiplier [1, MAX_FOURIER] float32, takes-cos mask, drop mask [n, MAX_FOURIER])


def _idft_rows(times: Sequence[float], dims: Sequence[int]) -> torch.Tensor:
    """IDFT(time_i, dim_i) for all models at once, [n, MAX_FOURIER] float32 (zero beyond dim_i): the
    reference's formula (sh_utils.py:120-130) applied to a column of times -- the same float32
    elementwise operations in the same order ((pi t) k, then cos / sin), hence the same bits as one call per
    model (tests/test_compose.py pins both against the reference-derived fixture), at a fraction of the host time:
    a per-actor call costs 50 us of Python / dispatcher overhead, and everything that depends on the models only
    (the multipliers as float32 -- exact small integers, what the reference's int64 -> float32 promotion yields --,
    which columns take the cosine, which columns a model does not use) is built once per set of Fourier dimensions.
    This runs once per FRAME in the simulator's loop, in front of the frame's first launch (the GPU waits for it):
    seven small tensor operations (round 6: 60 -> 25 -> 17 us)."""
    key = dims if isinstance(dims, tuple) else tuple(int(d) for d in dims)
    c = _IDFT_CONST.get(key)
    if c is None:
        indices = torch.arange(MAX_FOURIER)
        is_even = (indices % 2 == 0).view(1, -1)
        mult = torch.where(is_even, indices.view(1, -1), indices.view(1, -1) + 1).float()   # k (even k), k + 1 (odd k)
        keep = indices.view(1, -1) < torch.tensor(key).view(-1, 1)
        c = _IDFT_CONST[key] = (mult, is_even.expand(len(key), -1).contiguous(), ~keep)
    mult, takes_cos, drop = c
    t = torch.from_numpy(np.asarray(times, dtype=np.float32)).view(-1, 1)
    arg = (torch.pi * t) * mult                     # (pi t) k in float32, like torch.pi * t * even / (odd + 1)
    return torch.where(takes_cos, torch.cos(arg), torch.sin(arg)).masked_fill_(drop, 0.0)


def _model_lists(models: Sequence[ModelParams]):
    dims = []
    for m in models:
        F = int(m.features_dc.shape[1])
        if F > MAX_FOURIER:
            raise ValueError("fourier_dim %d > %d" % (F, MAX_FOURIER))
        dims.append(F)
    lists = [[m[f] for m in models] for f in range(6)]
    lists.append([_NO_FLIP if m.flip is None else m.flip for m in models])
    return lists, tuple(dims)


def _pack(models: Sequence[ModelParams], poses: Sequence[Optional[ActorPose]]):
    if len(models) != len(poses):
        raise ValueError("one pose entry (None for a static model) per model")
    lists, dims = _model_lists(models)
    rows, times = [], []
    tens, slots = [], []   # tensor-valued pose parts and where they go: ONE host copy for all of them
    for i, p in enumerate(poses):
        if p is None:      # a static model has fourier_dim 1: weight cos(0) = 1
            rows.append(_ZERO_ROW)
            times.append(0.0)
            continue
        row = [1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
        for off, v, n in ((1, p.obj_rot, 4), (5, p.obj_trans, 3)):
            if isinstance(v, torch.Tensor):
                if v.numel() != n:
                    raise ValueError("pose component of %d elements, expected %d" % (v.numel(), n))
                tens.append(v.detach().reshape(-1).float())
                slots.append((i, off, n))
            else:
                if len(v) != n:
                    raise ValueError("pose component of %d elements, expected %d" % (len(v), n))
                row[off:off + n] = [float(x) for x in v]
        rows.append(row)
        times.append(p.fourier_time)
    if tens:
        # tracked poses usually live on the GPU: a .cpu() per actor would be a device sync per actor and
        # frame; stack them first (one small kernel) and copy once
        if len({t.device for t in tens}) == 1:
            flat_vals = torch.cat(tens).cpu().tolist()
        else:
            flat_vals = torch.cat([t.cpu() for t in tens]).tolist()
        k = 0
        for i, off, n in slots:
            rows[i][off:off + n] = flat_vals[k:k + n]
            k += n
    pose_t = torch.from_numpy(np.asarray(rows, dtype=np.float32))     # [n, 8] (float64 -> float32 like torch.tensor)
    idft_t = _idft_rows(times, dims)
    return lists, pose_t, idft_t


def compose(models: Sequence[ModelParams], poses: Sequence[Optional[ActorPose]]):
    """The composition alone: (means3D [P,3], scales [P,3], rotations [P,4], opacity [P,1],
    shs [P,M,3]) -- what get_xyz / get_scaling / get_rotation / get_opacity / get_features of the
    reference return -- computed by the same HIP code the fused rasterizer uses."""
    lists, pose_t, idft_t = _pack(models, poses)
    return _C.compose(*lists, pose_t, idft_t)


class _ComposedRasterize(torch.autograd.Function):
    """Training path: forward = _C.rasterize_gaussians_composed(for_backward=True), backward =
    _C.rasterize_gaussians_composed_backward (C ABI grpg_backward_composed)."""

    @staticmethod
    def forward(ctx, owner, rs, nm, pose_t, idft_t, flips, pose_rot, pose_trans, means2D, *flat):
        lists = [list(flat[f * nm:(f + 1) * nm]) for f in range(6)]
        num_rendered, color, depth, alpha, radii, geom, binning, img = _C.rasterize_gaussians_composed(
            rs.bg, *lists, flips, pose_t, idft_t, rs.scale_modifier, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
            rs.tanfovy, rs.image_height, rs.image_width, rs.sh_degree, rs.campos, rs.debug, True)
        ctx.rs, ctx.nm, ctx.num_rendered = rs, nm, num_rendered
        owner.num_rendered = num_rendered   # like the evaluation branch of ComposedRasterizer.forward
        ctx.pose_t, ctx.idft_t, ctx.flips = pose_t, idft_t, flips
        ctx.pose_dev = (None if pose_rot is None else pose_rot.device,
                        None if pose_trans is None else pose_trans.device)
        ctx.save_for_backward(radii, alpha, geom, binning, img, *flat)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, g_color, g_radii, g_depth, g_alpha):
        rs, nm = ctx.rs, ctx.nm
        radii, alpha, geom, binning, img = ctx.saved_tensors[:5]
        flat = ctx.saved_tensors[5:]
        lists = [list(flat[f * nm:(f + 1) * nm]) for f in range(6)]
        zeros = lambda t: torch.zeros_like(t)   # noqa: E731  (an output the loss does not touch)
        gx, gs, gr, go, gdc, gfr, g_means2D, g_poses = _C.rasterize_gaussians_composed_backward(
            rs.bg, *lists, ctx.flips, ctx.pose_t, ctx.idft_t, rs.scale_modifier, rs.viewmatrix, rs.projmatrix,
            rs.tanfovx, rs.tanfovy, rs.sh_degree, rs.campos, radii, alpha, geom, ctx.num_rendered,
            binning, img, g_color if g_color is not None else zeros(alpha).expand(3, -1, -1).contiguous(),
            g_depth if g_depth is not None else zeros(alpha),
            g_alpha if g_alpha is not None else zeros(alpha), rs.debug)
        need = ctx.needs_input_grad
        g_rot = g_poses[:, 0:4].to(ctx.pose_dev[0]) if need[6] else None
        g_trans = g_poses[:, 4:7].to(ctx.pose_dev[1]) if need[7] else None
        grads = [g for per_field in (gx, gs, gr, go, gdc, gfr) for g in per_field]
        flat_grads = tuple(g if n else None for g, n in zip(grads, need[9:]))
        return (None, None, None, None, None, None, g_rot, g_trans, g_means2D if need[8] else None) + flat_grads


class ComposedRasterizer(nn.Module):
    """``GaussianRasterizer`` for a scene graph: takes per-model raw parameters + per-frame actor
    poses, returns ``(color, radii, depth, alpha)`` like the classic module (no semantics).
    Differentiable with respect to the raw parameters, tensor-valued poses and ``means2D``."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward_layers(self, models: Sequence[ModelParams], poses: Sequence[Optional[ActorPose]],
                       object_models: Optional[Sequence[bool]] = None, layer_bg=None) -> dict:
        """Evaluation only: the composition AND the three renders of the reference's
        ``StreetGaussianRenderer.render_all`` (street_gaussian_renderer.py:13-40: all models, the background
        model alone, the object models alone -- the last two on ``layer_bg``, default white) from ONE call on the
        raw parameters (C ABI ``grpg_forward_composed_layers``).  ``object_models``: one flag per model, default
        "every posed model (actor) is an object".  Returns the dict of ``GaussianRasterizer.forward_layers``;
        every plane is bit-identical to that call on ``compose(models, poses)``'s output."""
        rs = self.raster_settings
        lists, pose_t, idft_t = _pack(models, poses)
        dev = models[0].xyz.device
        if layer_bg is None:
            layer_bg = torch.ones(3, dtype=torch.float32, device=dev)
        flags = torch.empty(0, dtype=torch.uint8) if object_models is None else \
            torch.tensor([1 if f else 0 for f in object_models], dtype=torch.uint8)
        with torch.no_grad():
            (num_rendered, color, depth, alpha, radii, color_bg, alpha_bg, color_obj,
             alpha_obj) = _C.rasterize_gaussians_composed_layers(
                rs.bg, layer_bg, flags, *lists, pose_t, idft_t, rs.scale_modifier, rs.viewmatrix, rs.projmatrix,
                rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, rs.sh_degree, rs.campos, rs.debug)
        self.num_rendered = num_rendered
        return {"color": color, "radii": radii, "depth": depth, "alpha": alpha,
                "color_background": color_bg, "alpha_background": alpha_bg,
                "color_object": color_obj, "alpha_object": alpha_obj}

    def forward_frame(self, models: Sequence[ModelParams], poses: Sequence[Optional[ActorPose]], *,
                      sky_cube=None, ray_matrix=None, sky_fill=0.0, clamp=True, planes=False, rgb8=True,
                      truncate=True, out=None, layers=False, object_models: Optional[Sequence[bool]] = None,
                      layer_bg=None) -> dict:
        """The reference's whole per-frame evaluation path as ONE call on the raw parameters (C ABI
        ``grpg_forward_composed_frame``): scene-graph composition (street_gaussian_model.py:296-453), the op,
        ``StreetGaussianRenderer.render``'s clamp / sky composite / clamp (street_gaussian_renderer.py:106-116,
        236-237) and the simulator's uint8 [H,W,3] conversion (simulator.py:313-314) -- the last three inside the
        render's epilogue.  ``layers=True`` also returns render_all's two layer renders (forward_layers); the
        simulator reads ``result['rgb']`` only and does not need them.  Returns a dict: ``rgb8`` (uint8 [H,W,3] on the
        device; ``out``: a preallocated device tensor, or a PINNED HOST tensor -- the bytes then land in host memory
        while the render runs, no copy behind the launch: what the simulator wants; synchronise with the stream before
        reading), with ``planes=True`` also ``rgb`` (the FINAL colour) / ``depth`` /
        ``alpha``, with ``layers=True`` the four layer planes; ``radii``, ``num_rendered``."""
        rs = self.raster_settings
        lists, pose_t, idft_t = _pack(models, poses)
        dev = models[0].xyz.device
        e = torch.Tensor([])
        if layers and layer_bg is None:
            layer_bg = torch.ones(3, dtype=torch.float32, device=dev)
        flags = torch.empty(0, dtype=torch.uint8) if object_models is None else \
            torch.tensor([1 if f else 0 for f in object_models], dtype=torch.uint8)
        with torch.no_grad():
            (n, frame, color, depth, alpha, radii, color_bg, alpha_bg, color_obj,
             alpha_obj) = _C.rasterize_gaussians_composed_frame(
                rs.bg, e if layer_bg is None else layer_bg, flags, bool(layers), *lists, pose_t, idft_t,
                rs.scale_modifier, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height,
                rs.image_width, rs.sh_degree, rs.campos, rs.debug, e if sky_cube is None else sky_cube,
                e if ray_matrix is None else ray_matrix, float(sky_fill), bool(clamp), bool(planes), bool(rgb8),
                bool(truncate), out)
        self.num_rendered = n
        res = {"num_rendered": n, "radii": radii}
        if rgb8:
            res["rgb8"] = frame
        if planes:
            res.update(rgb=color, depth=depth, alpha=alpha)
        if layers:
            res.update(color_background=color_bg, alpha_background=alpha_bg, color_object=color_obj,
                       alpha_object=alpha_obj)
        return res

    def forward(self, models: Sequence[ModelParams], poses: Sequence[Optional[ActorPose]], means2D=None):
        rs = self.raster_settings
        lists, pose_t, idft_t = _pack(models, poses)
        flips = lists[6]
        flat = [t for per_field in lists[:6] for t in per_field]
        pose_tensors = [t for p in poses if p is not None for t in (p.obj_rot, p.obj_trans)
                        if isinstance(t, torch.Tensor)]
        train = torch.is_grad_enabled() and any(
            t.requires_grad for t in flat + pose_tensors + ([means2D] if means2D is not None else []))
        if not train:
            with torch.no_grad():
                num_rendered, color, depth, alpha, radii, geom, binning, img = _C.rasterize_gaussians_composed(
                    rs.bg, *lists, pose_t, idft_t, rs.scale_modifier, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
                    rs.tanfovy, rs.image_height, rs.image_width, rs.sh_degree, rs.campos, rs.debug)
            self.num_rendered = num_rendered
            return color, radii, depth, alpha
        # poses as [n,4] / [n,3] tensors that keep their graph (rows of static models: constants)
        pose_rot = pose_trans = None
        if pose_tensors:
            one = torch.tensor([1.0, 0.0, 0.0, 0.0])
            zero3 = torch.zeros(3)
            dev = pose_tensors[0].device
            as_t = lambda v, d: v.to(dev).float() if isinstance(v, torch.Tensor) else \
                torch.tensor([float(x) for x in v], device=dev)   # noqa: E731
            pose_rot = torch.stack([one.to(dev) if p is None else as_t(p.obj_rot, dev) for p in poses])
            pose_trans = torch.stack([zero3.to(dev) if p is None else as_t(p.obj_trans, dev) for p in poses])
        return _ComposedRasterize.apply(self, rs, len(models), pose_t, idft_t, flips, pose_rot, pose_trans,
                                        means2D, *flat)
