"""CPU tests (-m "not gpu"): gaussianrpg_amd/checkpoint.py reads the reference's trained-model files
-- the .pth state dict of StreetGaussianModel.save_state_dict (R/lib/models/street_gaussian_model.py:
138-158, gaussian_model.py:182-205) and the per-model PLY elements of save_ply (:94-105, gaussian_model.py:
84-96) -- into the raw per-model parameters, and applies the reference's activations
(gaussian_model.py:208-251).  The files are written here in the reference's layout; the PLY is also
decoded a second time with the arithmetic of GaussianModel.load_ply restated in the test."""
import os

import numpy as np
import pytest
import torch

from gaussianrpg_amd import checkpoint as ckpt
from gaussianrpg_amd.composed import ModelParams


def _model(n, F, M, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)      # noqa: E731
    return ModelParams(r(n, 3) * 5, r(n, 3) - 2.0, r(n, 4), r(n, 1), r(n, F, 3), r(n, M - 1, 3) * 0.2)


@pytest.fixture()
def models():
    return {"background": _model(500, 1, 4, 1), "obj_007": _model(60, 3, 4, 2), "obj_012": _model(40, 3, 4, 3)}


def _same(loaded, models):
    assert loaded.names() == list(models)
    for name, m in models.items():
        got = loaded.params(name)
        for f in ModelParams._fields[:6]:
            assert torch.equal(getattr(got, f), getattr(m, f).float()), (name, f)


def test_pth_state_dict_layout(tmp_path, models):
    sd = ckpt.state_dict_of(models, semantic_classes=5)
    # what a non-final checkpoint of the reference carries besides the model tensors
    for name in sd:
        sd[name] = {k: torch.nn.Parameter(v) if k != "semantic" else v for k, v in sd[name].items()}
        sd[name].update(spatial_lr_scale=1.5, denom=torch.zeros(3, 1), active_sh_degree=1,
                        max_radii2D=torch.zeros(3))
    sd["actor_pose"] = {"params": {"opt_trans": torch.zeros(4, 2, 3)}}
    sd["sky_cubemap"] = {"params": {"sky_cube_map": torch.zeros(6, 8, 8, 3)}}
    path = str(tmp_path / "iteration_30000.pth")
    torch.save(sd, path)
    loaded = ckpt.load_checkpoint(path)
    _same(loaded, models)
    assert loaded.sh_degree("background") == 1 and loaded.semantic("obj_007").shape == (60, 5)
    assert loaded.num_gaussians() == 600 and loaded.num_gaussians(["background"]) == 500
    # a bare GaussianModel.state_dict() is one model called background
    path2 = str(tmp_path / "single.pth")
    torch.save(ckpt.state_dict_of({"background": models["background"]})["background"], path2)
    _same(ckpt.load_checkpoint(path2), {"background": models["background"]})


def test_ply_layout_matches_the_references_load_ply(tmp_path, models):
    path = str(tmp_path / "point_cloud.ply")
    ckpt.write_ply(path, models, semantic_classes=2)
    _same(ckpt.load_checkpoint(path), models)
    # decode the same file with GaussianModel.load_ply's own steps (gaussian_model.py:113-133)
    el = ckpt.read_ply(path)["vertex_obj_007"]
    names = el.dtype.names
    assert names[:6] == ("x", "y", "z", "nx", "ny", "nz") and names[-2:] == ("semantic_0", "semantic_1")
    dc_names = sorted((n for n in names if n.startswith("f_dc_")), key=lambda x: int(x.split("_")[-1]))
    features_dc = np.zeros((el.shape[0], len(dc_names)))
    for idx, attr in enumerate(dc_names):
        features_dc[:, idx] = np.asarray(el[attr])
    features_dc = features_dc.reshape(features_dc.shape[0], 3, -1)
    ref = torch.tensor(features_dc, dtype=torch.float).transpose(1, 2).contiguous()
    assert torch.equal(ref, models["obj_007"].features_dc)


def test_ascii_ply_and_bad_files(tmp_path, models):
    m = models["background"]
    n = 7
    cols = torch.cat([m.xyz[:n], torch.zeros(n, 3), m.features_dc[:n].transpose(1, 2).flatten(1),
                      m.features_rest[:n].transpose(1, 2).flatten(1), m.opacity[:n], m.scaling[:n],
                      m.rotation[:n]], dim=1).numpy()
    props = (["x", "y", "z", "nx", "ny", "nz"] + ["f_dc_%d" % i for i in range(3)] +
             ["f_rest_%d" % i for i in range(9)] + ["opacity"] + ["scale_%d" % i for i in range(3)] +
             ["rot_%d" % i for i in range(4)])
    path = str(tmp_path / "a.ply")
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment made by a test\nelement vertex %d\n" % n)
        f.write("".join("property float %s\n" % p for p in props) + "end_header\n")
        for row in cols:
            f.write(" ".join(repr(float(v)) for v in row) + "\n")
    got = ckpt.load_checkpoint(path).params("background")
    assert torch.allclose(got.xyz, m.xyz[:n]) and torch.allclose(got.features_rest, m.features_rest[:n])
    bad = str(tmp_path / "bad.pth")
    torch.save({"something": torch.zeros(3)}, bad)
    with pytest.raises(ValueError, match="no Gaussian model"):
        ckpt.load_checkpoint(bad)
    wrong = ckpt.state_dict_of({"background": m})
    wrong["background"]["rotation"] = torch.zeros(500, 3)
    torch.save(wrong, bad)
    with pytest.raises(ValueError, match="rotation has shape"):
        ckpt.load_checkpoint(bad)


def test_activated_scene_is_what_the_references_getters_return(tmp_path, models):
    path = str(tmp_path / "m.pth")
    torch.save(ckpt.state_dict_of(models), path)
    loaded = ckpt.load_checkpoint(path)
    sc = ckpt.activated_scene(loaded)                 # static models only: the background
    m = models["background"]
    assert sc.means3D.shape == (500, 3) and sc.sh_degree == 1 and sc.shs.shape == (500, 4, 3)
    assert torch.equal(sc.scales, torch.exp(m.scaling))                                 # get_scaling
    assert torch.equal(sc.rotations, torch.nn.functional.normalize(m.rotation))          # get_rotation
    assert torch.equal(sc.opacity, torch.sigmoid(m.opacity))                            # get_opacity
    assert torch.equal(sc.shs, torch.cat((m.features_dc, m.features_rest), dim=1))       # get_features
    with pytest.raises(ValueError, match="fourier_dim"):
        ckpt.activated_scene(loaded, ["obj_007"])
    # actors join through the composed rasterizer, and only with a pose
    from gaussianrpg_amd.composed import ActorPose
    ms, ps = ckpt.scene_models(loaded, "cpu", poses={"obj_012": ActorPose([1, 0, 0, 0], [2, 0, 5], 0.3)})
    assert len(ms) == 2 and ps[0] is None and ps[1].obj_trans == [2, 0, 5]
    assert torch.equal(ms[1].xyz, models["obj_012"].xyz)


def _write_ply_table(path, element, names, table):
    """A PLY container around an element table, independent of checkpoint.write_ply."""
    hdr = ["ply", "format binary_little_endian 1.0", "element %s %d" % (element, table.shape[0])]
    hdr += ["property float %s" % n for n in names] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(hdr) + "\n").encode("ascii"))
        f.write(np.ascontiguousarray(table, dtype="<f4").tobytes())


def test_loader_inverts_the_references_own_writer(tmp_path):
    """ref_ply_layout.npz (tests/golden/make_golden.py part_a_ply_layout): the element table and the
    property names GaussianModel.make_ply produced IN the reference's code for seven known tensors, and
    the keys of its state_dict(is_final=True).  The loader must give those tensors back -- bit for bit,
    feature planes in the [N, K, 3] layout of the state dict (make_ply stores them channel-major)."""
    from helpers import GOLDEN
    z = np.load(os.path.join(GOLDEN, "ref_ply_layout.npz"))
    names = [str(n) for n in z["property_names"]]
    want = {k: z["sd_" + k] for k in (str(s) for s in z["state_dict_keys"])}
    assert list(want) == ["xyz", "feature_dc", "feature_rest", "scaling", "rotation", "opacity", "semantic"]
    # PLY: one element of a street scene (save_ply names it vertex_<model>, street_gaussian_model.py:94-105)
    ply = str(tmp_path / "point_cloud.ply")
    _write_ply_table(ply, "vertex_background", names, z["table"])
    got = ckpt.load_checkpoint(ply)
    assert got.names() == ["background"] and got.sh_degree("background") == 3
    for k, v in want.items():
        assert np.array_equal(got.models["background"][k].numpy(), v), k
    # .pth: the state dict under the reference's keys, as save_state_dict nests it per model
    pth = str(tmp_path / "iteration_7000.pth")
    torch.save({"background": {k: torch.tensor(v) for k, v in want.items()}}, pth)
    got = ckpt.load_checkpoint(pth)
    for k, v in want.items():
        assert np.array_equal(got.models["background"][k].numpy(), v), k
    # and the writer of this package produces the reference's table and property order
    out = str(tmp_path / "rewritten.ply")
    m = got.params("background")
    ckpt.write_ply(out, {"background": m}, semantic_classes=want["semantic"].shape[1])
    el = ckpt.read_ply(out)["vertex_background"]
    assert list(el.dtype.names) == names
    sem_cols = [i for i, n in enumerate(names) if n.startswith("semantic_")]
    keep = [i for i in range(len(names)) if i not in sem_cols]     # write_ply has no semantic values to write
    table = np.stack([el[n] for n in names], axis=1)
    assert np.array_equal(table[:, keep], z["table"][:, keep])


class _NeedsFullUnpickler:          # what a non-final checkpoint's optimizer / bidict objects look like to
    def __init__(self):             # torch.load(weights_only=True): a class it does not know
        self.x = 1


def test_unsafe_unpickler_only_on_request(tmp_path, models):
    """a .pth that weights_only=True cannot read is refused unless the caller opts in."""
    path = str(tmp_path / "nonfinal.pth")
    sd = ckpt.state_dict_of(models)
    sd["optimizer_like"] = _NeedsFullUnpickler()
    torch.save(sd, path)
    with pytest.raises(ValueError, match="allow_unsafe=True"):
        ckpt.load_checkpoint(path)
    with pytest.warns(UserWarning, match="unsafe"):
        got = ckpt.load_checkpoint(path, allow_unsafe=True)
    assert set(got.names()) == set(models)
