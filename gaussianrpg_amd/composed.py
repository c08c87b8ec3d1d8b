"""Fused scene-graph composition in front of the rasterizer (additive API).

Every frame the reference's ``StreetGaussianModel`` rebuilds the op's flat inputs in PyTorch
(lib/models/street_gaussian_model.py:296-453): per model ``exp`` / ``sigmoid`` / ``normalize`` of the
raw parameters (gaussian_model.py:224-251), per visible actor a rigid transform of the means and a
quaternion product for the rotations with the tracked pose (street_gaussian_model.py:318-365), an
inverse DFT of the actor's Fourier colour coefficients (gaussian_model_actor.py:73-82) and a
``torch.cat`` over the models -- about twenty small kernels and two extra passes over the op's whole
input.  ``ComposedRasterizer`` hands the models' RAW parameter tensors and the per-frame poses to
``_C.rasterize_gaussians_composed`` instead; the arithmetic happens inside the HIP preprocess and the
concatenated tensors are never materialised.  Forward only (evaluation, trajectory and simulator
rendering); training keeps composing in PyTorch, where autograd needs the intermediates, and calls
the classic ``GaussianRasterizer``.

    models = [ModelParams(...background...), ModelParams(...actor 1...), ...]
    poses  = [None, ActorPose(obj_rot, obj_trans, fourier_time), ...]
    color, radii, depth, alpha = ComposedRasterizer(raster_settings)(models, poses)
"""
import math
from typing import List, NamedTuple, Optional, Sequence

import torch
import torch.nn as nn

from .rasterizer import GaussianRasterizationSettings, _C

MAX_FOURIER = 8     # GRPG_MAX_FOURIER


class ModelParams(NamedTuple):
    """Raw (pre-activation) parameters of one Gaussian model, as the reference stores them
    (gaussian_model.py:36-47): ``_xyz [N,3]``, ``_scaling [N,3]`` (log), ``_rotation [N,4]``,
    ``_opacity [N,1]`` (logit), ``_features_dc [N,F,3]`` (F = fourier_dim, 1 for the background),
    ``_features_rest [N,M-1,3]``."""
    xyz: torch.Tensor
    scaling: torch.Tensor
    rotation: torch.Tensor
    opacity: torch.Tensor
    features_dc: torch.Tensor
    features_rest: torch.Tensor


class ActorPose(NamedTuple):
    """Pose of one actor for one frame: ``obj_rot`` (w,x,y,z) and ``obj_trans`` in world space, ego
    pose already applied (street_gaussian_model.py:268-273); ``fourier_time`` =
    fourier_scale * (frame - start_frame) / (end_frame - start_frame) (gaussian_model_actor.py:74-75)."""
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


def _idft_rows(times: Sequence[float], dims: Sequence[int]) -> torch.Tensor:
    """IDFT(time_i, dim_i) for all models at once, [n, MAX_FOURIER] float32 (zero beyond dim_i): the
    reference's formula (sh_utils.py:120-130) applied to a column of times -- the same float32
    elementwise operations, hence the same bits as one call per model, at one fiftieth of the host
    time (a per-actor call costs 50 us of Python / dispatcher overhead)."""
    n = len(times)
    t = torch.tensor([float(x) for x in times], dtype=torch.float32).view(-1, 1)
    idft = torch.zeros(n, MAX_FOURIER)
    indices = torch.arange(MAX_FOURIER)
    even, odd = indices[::2], indices[1::2]
    idft[:, even] = torch.cos(torch.pi * t * even)
    idft[:, odd] = torch.sin(torch.pi * t * (odd + 1))
    keep = indices.view(1, -1) < torch.tensor([int(d) for d in dims]).view(-1, 1)
    return torch.where(keep, idft, torch.zeros(()))


def _pack(models: Sequence[ModelParams], poses: Sequence[Optional[ActorPose]]):
    if len(models) != len(poses):
        raise ValueError("one pose entry (None for a static model) per model")
    rows, times, dims = [], [], []
    for m, p in zip(models, poses):
        F = int(m.features_dc.shape[1])
        if F > MAX_FOURIER:
            raise ValueError("fourier_dim %d > %d" % (F, MAX_FOURIER))
        dims.append(F)
        if p is None:      # a static model has fourier_dim 1: weight cos(0) = 1
            rows.append([0.0] * 8)
            times.append(0.0)
        else:
            rows.append([1.0] + [float(v) for v in p.obj_rot] + [float(v) for v in p.obj_trans])
            times.append(float(p.fourier_time))
    pose_t = torch.tensor(rows, dtype=torch.float32).reshape(len(models), 8)
    idft_t = _idft_rows(times, dims)
    lists = [[getattr(m, f) for m in models] for f in ModelParams._fields]
    return lists, pose_t, idft_t


def compose(models: Sequence[ModelParams], poses: Sequence[Optional[ActorPose]]):
    """The composition alone: (means3D [P,3], scales [P,3], rotations [P,4], opacity [P,1],
    shs [P,M,3]) -- what get_xyz / get_scaling / get_rotation / get_opacity / get_features of the
    reference return -- computed by the same HIP code the fused rasterizer uses."""
    lists, pose_t, idft_t = _pack(models, poses)
    return _C.compose(*lists, pose_t, idft_t)


class ComposedRasterizer(nn.Module):
    """``GaussianRasterizer`` for a scene graph: takes per-model raw parameters + per-frame actor
    poses, returns ``(color, radii, depth, alpha)`` like the classic module (no semantics)."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    @torch.no_grad()
    def forward(self, models: Sequence[ModelParams], poses: Sequence[Optional[ActorPose]]):
        rs = self.raster_settings
        lists, pose_t, idft_t = _pack(models, poses)
        num_rendered, color, depth, alpha, radii, geom, binning, img = _C.rasterize_gaussians_composed(
            rs.bg, *lists, pose_t, idft_t, rs.scale_modifier, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
            rs.tanfovy, rs.image_height, rs.image_width, rs.sh_degree, rs.campos, rs.debug)
        self.num_rendered = num_rendered
        return color, radii, depth, alpha
