"""Readers for the reference's trained-model files, so that the harness / bench run on REAL scenes.

The reference stores a Street-Gaussians scene in two forms (both per model: ``background``,
``obj_<track id>``, optionally ``sky``):

* ``trained_model/iteration_<n>.pth`` -- ``torch.save`` of ``StreetGaussianModel.save_state_dict``
  (R/lib/models/street_gaussian_model.py:138-158): ``{model_name: GaussianModel.state_dict(), ...,
  'actor_pose': ..., 'sky_cubemap': ..., ...}`` with, per model (R/lib/models/gaussian_model.py:182-205),
  ``xyz [N,3]``, ``feature_dc [N,F,3]`` (F = fourier_dim of an actor, 1 otherwise), ``feature_rest
  [N,M-1,3]``, ``scaling [N,3]`` (log), ``rotation [N,4]`` (raw quaternion r,x,y,z), ``opacity [N,1]``
  (logit), ``semantic [N,C]`` -- the RAW parameters, before the activations;
* ``point_cloud/iteration_<n>/point_cloud.ply`` -- one PLY element ``vertex_<model_name>`` per model
  (R/lib/models/street_gaussian_model.py:94-105) with the float properties ``x y z nx ny nz f_dc_*
  f_rest_* opacity scale_* rot_* semantic_*`` (R/lib/models/gaussian_model.py:84-96); ``f_dc_k`` /
  ``f_rest_k`` are the ``[N,3,K]`` (channel-major) flattening of the feature tensors (:86-87,125-126).

This module reads both with numpy / torch only (``plyfile`` and ``lib.*`` are not needed) into
``composed.ModelParams`` -- the raw layout ``ComposedRasterizer`` consumes -- and can apply the
reference's activations (gaussian_model.py:208-251) to hand the classic flat op a ``harness.Scene``.
Actor poses are not part of either file (they come from the dataset's tracklets,
R/lib/models/actor_pose.py); ``scene_from_models`` therefore renders static models as they are and
actors only when the caller provides their poses.
"""
import math
import os
from collections import OrderedDict
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from .composed import ActorPose, ModelParams

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
              "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
              "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def read_ply(path: str) -> "OrderedDict[str, np.ndarray]":
    """Minimal PLY reader: ``{element name: structured array}`` for files made of fixed-size scalar
    properties (what the reference writes); ascii, binary_little_endian and binary_big_endian."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s is not a PLY file" % path)
        fmt, elements = None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: unterminated PLY header" % path)
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append((tok[1], int(tok[2]), []))
            elif tok[0] == "property":
                if tok[1] == "list":
                    raise ValueError("%s: list properties are not supported" % path)
                elements[-1][2].append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        out = OrderedDict()
        for name, count, props in elements:
            if fmt == "ascii":
                rows = [f.readline().split() for _ in range(count)]
                arr = np.zeros(count, dtype=[(p, t) for p, t in props])
                for j, (p, t) in enumerate(props):
                    arr[p] = np.array([r[j] for r in rows], dtype=np.float64).astype(t)
            else:
                order = "<" if fmt == "binary_little_endian" else ">"
                dt = np.dtype([(p, order + t) for p, t in props])
                arr = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
            out[name] = arr
    return out


def _numbered(names: Sequence[str], prefix: str):
    return sorted((n for n in names if n.startswith(prefix)), key=lambda n: int(n.split("_")[-1]))


def _model_from_ply_element(el: np.ndarray) -> Dict[str, torch.Tensor]:
    """The arithmetic of GaussianModel.load_ply (R/lib/models/gaussian_model.py:104-155)."""
    names = el.dtype.names
    n = el.shape[0]
    col = lambda k: np.asarray(el[k], dtype=np.float32)                                   # noqa: E731
    stack = lambda ks: (np.stack([col(k) for k in ks], axis=1) if ks else np.zeros((n, 0), np.float32))  # noqa: E731
    f_dc = stack(_numbered(names, "f_dc_")).reshape(n, 3, -1)          # [N,3,F] channel-major
    f_rest = stack(_numbered(names, "f_rest_")).reshape(n, 3, -1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))                                # noqa: E731
    return {"xyz": t(stack(["x", "y", "z"])), "feature_dc": t(f_dc.transpose(0, 2, 1)),
            "feature_rest": t(f_rest.transpose(0, 2, 1)), "scaling": t(stack(_numbered(names, "scale_"))),
            "rotation": t(stack(_numbered(names, "rot_"))), "opacity": t(col("opacity")[:, None]),
            "semantic": t(stack(_numbered(names, "semantic_")))}


class LoadedScene:
    """Raw per-model parameters of a checkpoint, in file order."""

    def __init__(self, models: "OrderedDict[str, Dict[str, torch.Tensor]]", source: str):
        self.models, self.source = models, source

    def names(self):
        return list(self.models)

    def num_gaussians(self, names: Optional[Sequence[str]] = None) -> int:
        return sum(int(self.models[n]["xyz"].shape[0]) for n in (names or self.models))

    def sh_degree(self, name: str) -> int:
        m = self.models[name]
        M = int(m["feature_rest"].shape[1]) + 1
        d = int(round(math.sqrt(M))) - 1
        if (d + 1) ** 2 != M:
            raise ValueError("%s: %d SH coefficients are not (degree + 1)^2" % (name, M))
        return d

    def params(self, name: str, device="cpu") -> ModelParams:
        m = self.models[name]
        f = lambda k: m[k].detach().to(device=device, dtype=torch.float32).contiguous()        # noqa: E731
        return ModelParams(f("xyz"), f("scaling"), f("rotation"), f("opacity"), f("feature_dc"), f("feature_rest"))

    def semantic(self, name: str, device="cpu") -> torch.Tensor:
        return self.models[name]["semantic"].detach().to(device=device, dtype=torch.float32).contiguous()


_MODEL_KEYS = ("xyz", "feature_dc", "feature_rest", "scaling", "rotation", "opacity")


def load_checkpoint(path: str, allow_unsafe: bool = False) -> LoadedScene:
    """``.pth`` (StreetGaussianModel.save_state_dict, or a bare GaussianModel.state_dict) or ``.ply``
    (one ``vertex_<model>`` element per model, or a single ``vertex`` element) -> LoadedScene.

    A ``.pth`` is read with ``torch.load(weights_only=True)``.  A non-final checkpoint of the
    reference (optimizer state, bidict objects) needs the full unpickler, which EXECUTES code found
    in the file: that is only done when the caller says the file is trusted (``allow_unsafe=True``,
    ``bench.py --trust-checkpoint``), never as a silent fallback."""
    ext = os.path.splitext(path)[1].lower()
    models = OrderedDict()
    if ext == ".ply":
        for name, el in read_ply(path).items():
            if not name.startswith("vertex") or "opacity" not in (el.dtype.names or ()):
                continue
            models[name[7:] if name.startswith("vertex_") else "background"] = _model_from_ply_element(el)
    else:
        try:
            sd = torch.load(path, map_location="cpu", weights_only=True)
        except Exception as e:      # optimizer state / bidict objects of a non-final checkpoint
            if not allow_unsafe:
                raise ValueError(
                    "%s cannot be read with torch.load(weights_only=True) (%s: %s).  If it is a non-final "
                    "checkpoint of the reference and you TRUST the file, pass allow_unsafe=True "
                    "(bench.py --trust-checkpoint): the full unpickler executes code from the file."
                    % (path, type(e).__name__, str(e).splitlines()[0] if str(e) else "")) from e
            import warnings
            warnings.warn("loading %s with the full (unsafe) unpickler on the caller's request" % path)
            sd = torch.load(path, map_location="cpu", weights_only=False)
        if all(k in sd for k in _MODEL_KEYS):
            sd = {"background": sd}
        for name, m in sd.items():
            if isinstance(m, dict) and all(k in m for k in _MODEL_KEYS):
                entry = {k: torch.as_tensor(m[k]).detach().float() for k in _MODEL_KEYS}
                sem = m.get("semantic")
                entry["semantic"] = (torch.as_tensor(sem).detach().float() if sem is not None
                                     else torch.zeros(entry["xyz"].shape[0], 0))
                models[name] = entry
    if not models:
        raise ValueError("%s holds no Gaussian model (expected the reference's state dict or PLY layout)" % path)
    for name, m in models.items():
        n = m["xyz"].shape[0]
        shapes = {"xyz": (n, 3), "scaling": (n, 3), "rotation": (n, 4), "opacity": (n, 1)}
        for k, shp in shapes.items():
            if tuple(m[k].shape) != shp:
                raise ValueError("%s.%s has shape %s, expected %s" % (name, k, tuple(m[k].shape), shp))
        if m["feature_dc"].dim() != 3 or m["feature_dc"].shape[0] != n or m["feature_dc"].shape[2] != 3:
            raise ValueError("%s.feature_dc has shape %s, expected [N,F,3]" % (name, tuple(m["feature_dc"].shape)))
        if m["feature_rest"].dim() != 3 or m["feature_rest"].shape[0] != n or m["feature_rest"].shape[2] != 3:
            raise ValueError("%s.feature_rest has shape %s, expected [N,M-1,3]" % (name, tuple(m["feature_rest"].shape)))
    return LoadedScene(models, path)


def activated_scene(loaded: LoadedScene, names: Optional[Sequence[str]] = None):
    """The STATIC models (fourier_dim 1) of a checkpoint as one ``harness.Scene`` of activated, flat
    tensors -- what the reference's getters hand the rasterizer (gaussian_model.py:208-251:
    exp / normalize / sigmoid / cat(features_dc, features_rest)), in file order like
    ``StreetGaussianModel``'s concatenation (street_gaussian_model.py:296-453).  Actors need their
    per-frame pose: use ``scene_models`` + ``ComposedRasterizer`` for those."""
    from .harness import Scene
    names = list(names or [n for n in loaded.names() if loaded.models[n]["feature_dc"].shape[1] == 1
                           and n != "sky"])
    if not names:
        raise ValueError("no static model selected")
    deg = {loaded.sh_degree(n) for n in names}
    if len(deg) != 1:
        raise ValueError("models with different SH degrees cannot be concatenated: %s" % sorted(deg))
    parts = [loaded.models[n] for n in names]
    if any(p["feature_dc"].shape[1] != 1 for p in parts):
        raise ValueError("a model with fourier_dim > 1 is an actor: it needs a pose and a timestamp")
    cat = lambda k: torch.cat([p[k].float() for p in parts], dim=0)                          # noqa: E731
    shs = torch.cat([cat("feature_dc"), cat("feature_rest")], dim=1).contiguous()
    return Scene(cat("xyz").contiguous(), torch.sigmoid(cat("opacity")).contiguous(),
                 torch.exp(cat("scaling")).contiguous(),
                 torch.nn.functional.normalize(cat("rotation")).contiguous(), shs, deg.pop())


def scene_models(loaded: LoadedScene, device, poses: Optional[Dict[str, ActorPose]] = None,
                 names: Optional[Sequence[str]] = None):
    """(models, poses) for ``ComposedRasterizer.forward``: the raw parameters of the selected models on
    ``device``, static models with pose None, actors (``obj_*``) with the pose the caller provides
    (an actor without one is left out: its placement is unknown)."""
    poses = poses or {}
    ms, ps = [], []
    for n in (names or loaded.names()):
        if n == "sky":
            continue
        is_actor = n.startswith("obj_")
        if is_actor and n not in poses:
            continue
        ms.append(loaded.params(n, device))
        ps.append(poses.get(n) if is_actor else None)
    return ms, ps


# ---- writers (used by the tests and by anybody who wants to hand a synthetic scene to the
# reference's own loaders): the exact layouts described above ----
def state_dict_of(models: "Dict[str, ModelParams]", semantic_classes: int = 0) -> dict:
    sd = OrderedDict()
    for name, m in models.items():
        sd[name] = {"xyz": m.xyz, "feature_dc": m.features_dc, "feature_rest": m.features_rest,
                    "scaling": m.scaling, "rotation": m.rotation, "opacity": m.opacity,
                    "semantic": torch.zeros(m.xyz.shape[0], semantic_classes)}
    return sd


def write_ply(path: str, models: "Dict[str, ModelParams]", semantic_classes: int = 0) -> None:
    """binary_little_endian PLY with one ``vertex_<model>`` element per model, property order of
    GaussianModel.construct_list_of_attributes / make_ply (gaussian_model.py:70-96)."""
    header = ["ply", "format binary_little_endian 1.0"]
    blobs = []
    for name, m in models.items():
        n = int(m.xyz.shape[0])
        f_dc = m.features_dc.detach().float().transpose(1, 2).flatten(start_dim=1).cpu().numpy()
        f_rest = m.features_rest.detach().float().transpose(1, 2).flatten(start_dim=1).cpu().numpy()
        cols = [m.xyz.detach().float().cpu().numpy(), np.zeros((n, 3), np.float32), f_dc, f_rest,
                m.opacity.detach().float().cpu().numpy(), m.scaling.detach().float().cpu().numpy(),
                m.rotation.detach().float().cpu().numpy(), np.zeros((n, semantic_classes), np.float32)]
        props = (["x", "y", "z", "nx", "ny", "nz"] + ["f_dc_%d" % i for i in range(f_dc.shape[1])] +
                 ["f_rest_%d" % i for i in range(f_rest.shape[1])] + ["opacity"] +
                 ["scale_%d" % i for i in range(3)] + ["rot_%d" % i for i in range(4)] +
                 ["semantic_%d" % i for i in range(semantic_classes)])
        header.append("element vertex_%s %d" % (name, n))
        header += ["property float %s" % p for p in props]
        blobs.append(np.ascontiguousarray(np.concatenate(cols, axis=1), dtype="<f4").tobytes())
    header.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        for b in blobs:
            f.write(b)
