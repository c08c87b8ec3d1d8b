"""CPU test (-m "not gpu"): the Python -> _C boundary of gaussianrpg_amd/rasterizer.py is the reference's.

tests/golden/ref_binding_trace.json was recorded by importing the reference's OWN wrapper
(submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py) over a recording
stand-in for its compiled `_C` (helpers.BindingRecorder; make_golden.py part_a_binding_trace) and driving
it through a training forward + backward, markVisible and visible_filter.  Here the same driver runs this
package's wrapper over the same recorder: every `_C` call must carry the same object or value in every
argument position, the outputs must come back in the same order and every input must receive the same one
of `_C`'s nine gradients -- so the reference's wrapper and this one are interchangeable above
csrc/torch_binding.cpp (rasterize_points.h:35-99 as bound in ext.cpp)."""
import json
import os

import torch

from helpers import GOLDEN, BindingRecorder, binding_trace


class Recorder(BindingRecorder):
    """+ the two additive entry points of this package's binding: the backward that skips the reference's
    pure intermediates (same 26 arguments, same nine results) and the forward of calls no backward can
    follow (same 20 arguments, the same first six results)."""

    def rasterize_gaussians_backward_lean(self, *args):
        out = self.rasterize_gaussians_backward(*args)
        self.calls[-1][0] = "rasterize_gaussians_backward_lean"
        return out

    def rasterize_gaussians_eval(self, *args):
        out = self.rasterize_gaussians(*args)
        self.calls[-1][0] = "rasterize_gaussians_eval"
        return out


def test_wrapper_calls_the_binding_exactly_like_the_references_wrapper(monkeypatch):
    import gaussianrpg_amd.rasterizer as mine
    want = json.load(open(os.path.join(GOLDEN, "ref_binding_trace.json")))
    rec = Recorder()
    monkeypatch.setattr(mine, "_C", rec)
    got = binding_trace(mine, rec, backward_alias={"rasterize_gaussians_backward_lean": "rasterize_gaussians_backward"})
    assert sorted(got) == sorted(want)
    for scenario in want:
        for (gn, ga), (wn, wa) in zip(got[scenario]["calls"], want[scenario]["calls"]):
            assert gn == wn and ga == wa, (scenario, gn, wn, [(i, a, b) for i, (a, b) in enumerate(zip(ga, wa)) if a != b])
        assert len(got[scenario]["calls"]) == len(want[scenario]["calls"])
        for key in ("returned", "input_grads", "mark_visible_returns", "filter_returns"):
            assert got[scenario][key] == want[scenario][key], (scenario, key)


def test_evaluation_entry_point_gets_the_forward_arguments(monkeypatch):
    """Without any input that requires grad the wrapper takes `_C.rasterize_gaussians_eval`: the same 20
    arguments as the reference's forward call, the same five outputs in the same order."""
    import gaussianrpg_amd.rasterizer as mine
    want = json.load(open(os.path.join(GOLDEN, "ref_binding_trace.json")))["sh_scales_rotations"]
    rec = Recorder()
    monkeypatch.setattr(mine, "_C", rec)
    P, H, W, S = 5, 4, 6, 2
    rs = mine.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=0.5, tanfovy=0.25, bg=rec.tensor("rs.bg", 3), scale_modifier=0.75,
        viewmatrix=rec.tensor("rs.viewmatrix", 4, 4), projmatrix=rec.tensor("rs.projmatrix", 4, 4), sh_degree=1,
        campos=rec.tensor("rs.campos", 3), prefiltered=False, debug=False)
    t = lambda n, *s: rec.tensor("in." + n, *s)      # noqa: E731
    with torch.no_grad():
        outs = mine.GaussianRasterizer(rs)(
            means3D=t("means3D", P, 3), means2D=None, opacities=t("opacities", P, 1), shs=t("sh", P, 4, 3),
            scales=t("scales", P, 3), rotations=t("rotations", P, 4), semantics=t("semantics", P, S))
    assert rec.calls == [["rasterize_gaussians_eval", want["calls"][0][1]]]
    assert [rec.describe(o) for o in outs] == want["returned"]
