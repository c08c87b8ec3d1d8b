"""Shared helpers for the parity tests (tests/ may use oracle/; the product may not)."""
import os

import numpy as np
import torch

from gaussianrpg_amd import harness as hz

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Tolerance north_star states for floating point outputs: "forward RGB/depth within 1e-4 fp32".
ATOL = 1e-4
RTOL = 1e-4
# A pixel the oracle flags "fragile" had an accept/reject decision within a few ulp of its
# threshold (alpha~1/255, T~1e-4, power~0): an implementation whose exp() differs in the last
# bits may take the other branch.  One flipped contribution at the alpha threshold is worth at
# most alpha*T*c <= (1/255)(1+|ref|) ~= 4e-3 (1+|ref|); that is the bound a
# fragile pixel must meet (measured worst case over the GPU suite: 1.3e-3).
FRAGILE_ATOL = 4e-3
# North-star bar for the whole image, fragile pixels INCLUDED: at least 99.99 % of the pixels
# within 1e-4 (SURVEY.md section 7 budgets "a handful of pixels per frame" for threshold flips);
# images smaller than 30 k pixels may have 3 such pixels.
MAX_BEYOND_FRAC = 1e-4
MAX_BEYOND_MIN_PIXELS = 3


def oracle_kwargs(cam, sh_degree, bg=None, scale_modifier=1.0):
    kw = hz.settings_kwargs(cam, sh_degree, bg=bg, scale_modifier=scale_modifier)
    kw.pop("prefiltered")
    kw.pop("debug")
    return kw


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, "oracle_%s.npz" % name))
    return {k: z[k] for k in z.files}


def fixture_oracle_inputs(fx):
    return dict(image_height=int(fx["H"]), image_width=int(fx["W"]), tanfovx=float(fx["tanfovx"]),
                tanfovy=float(fx["tanfovy"]), bg=fx["bg"], scale_modifier=1.0,
                viewmatrix=fx["viewmatrix"], projmatrix=fx["projmatrix"],
                sh_degree=int(fx["sh_degree"]), campos=fx["campos"])


PARITY_STATS = []     # one record per compared image plane; dumped by conftest at session end


def assert_image_close(name, got, ref, fragile=None, atol=ATOL, rtol=RTOL, max_fragile_frac=0.1,
                       fragile_atol=None, max_beyond_frac=MAX_BEYOND_FRAC):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = np.abs(got - ref)
    tol = atol + rtol * np.abs(ref)
    # what the exemption really hides: the share of ALL pixels (fragile ones included) beyond 1e-4
    test_id = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]
    PARITY_STATS.append(dict(
        test=test_id, plane=name, pixels=int(err.size), beyond_1e4=int((err > tol).sum()),
        frac_beyond_1e4=float((err > tol).mean()) if err.size else 0.0,
        max_rel_err=float((err / (1.0 + np.abs(ref))).max()) if err.size else 0.0,
        fragile_frac=float(np.asarray(fragile, dtype=bool).mean()) if fragile is not None else None))
    if fragile is None:
        bad = err > tol
        assert not bad.any(), "%s: %d px beyond tol, max err %.3e" % (name, bad.sum(), err.max())
        return
    nbad = int((err > tol).sum())
    assert nbad <= max(MAX_BEYOND_MIN_PIXELS, max_beyond_frac * err.size), (
        "%s: %d of %d values beyond 1e-4 (%.2e of the image; bar %.0e)" % (
            name, nbad, err.size, nbad / err.size, max_beyond_frac))
    frag = np.broadcast_to(np.asarray(fragile, dtype=bool), ref.shape[-2:])
    bad = (err > tol) & ~frag[None]
    assert not bad.any(), "%s: %d non-fragile px beyond tol, max err %.3e" % (
        name, bad.sum(), (err * ~frag[None]).max())
    assert frag.mean() <= max_fragile_frac, "%s: fragile fraction %.4f" % (name, frag.mean())
    scale = 1.0 + np.abs(ref)
    fa = FRAGILE_ATOL if fragile_atol is None else fragile_atol
    assert ((err / scale) * frag[None]).max() <= fa, "%s: fragile px err %.3e (bound %.1e)" % (
        name, ((err / scale) * frag[None]).max(), fa)


def gaussians_fed_by_fragile_pixels(o, tile=16):
    """bool[P]: Gaussians the oracle evaluated at a threshold-FRAGILE pixel with a footprint that
    passes, or nearly passes, the alpha test there.  At such a pixel an implementation whose exp()
    differs in the last bits may legitimately take the other branch of one accept / reject decision;
    that changes this pixel's transmittance for every later splat, i.e. the pixel's share of the
    gradient of EVERY Gaussian blended there.  Gradient elements of these Gaussians are exempt from
    the element-wise bar (rtol 1e-3 / atol 1e-5) and are held to the array-level bars only."""
    frag = np.asarray(o["fragile"], dtype=bool)
    H, W = frag.shape
    gx = (W + tile - 1) // tile
    pl = np.asarray(o["point_list"]).astype(np.int64)
    rg = np.asarray(o["ranges"]).astype(np.int64).reshape(-1, 2)
    ncon = np.asarray(o["n_contrib"]).astype(np.int64).reshape(H, W)
    m2d = np.asarray(o["means2D"], dtype=np.float64)
    co = np.asarray(o["conic_opacity"], dtype=np.float64)
    fed = np.zeros(m2d.shape[0], dtype=bool)
    ys, xs = np.nonzero(frag)
    for y, x in zip(ys, xs):
        t = (y // tile) * gx + (x // tile)
        b, e = rg[t]
        e = min(e, b + ncon[y, x] + 2)        # + the entries a flipped termination would add
        if e <= b:
            continue
        ids = pl[b:e]
        dx = m2d[ids, 0] - x
        dy = m2d[ids, 1] - y
        power = -0.5 * (co[ids, 0] * dx * dx + co[ids, 2] * dy * dy) - co[ids, 1] * dx * dy
        alpha = np.minimum(0.99, co[ids, 3] * np.exp(np.minimum(power, 0.0)))
        fed[ids[(power <= 1e-3) & (alpha >= 0.5 / 255.0)]] = True
    return fed


def float64_truth_gradients(sc, okw, gc, gd, ga, gs=None, semantics=None, colors=None, cov=None):
    """Gradients of the test loss by float64 autograd through oracle/torch_splat.py: the exact
    gradient of the reference's formulas, up to accept / clamp decisions that fall differently in
    float64 (those show up identically against every float32 implementation).  Used where the HIP
    backward and gs_oracle.c disagree beyond the element-wise bar.  gs_oracle.c restates the
    reference: every pixel's walk starts from T_final = 1 - alphas[pix] (backward.cu:468) with alphas
    the float32 SUM of the blend weights, so on a nearly opaque pixel the cancellation leaves T_final --
    and every term of that pixel -- 1e-3 .. 1e-2 off (tools/experiments/tfinal_noise.py).  The HIP
    forward writes alpha = 1 - T (the same quantity by telescoping, rounded once; within the image
    bar of the sum), so its backward recovers T to 6e-8 absolute: measured 7-10x closer to this
    truth than the oracle on every contested array (tools/experiments/grad_truth.py)."""
    import torch
    from oracle import torch_splat as ts
    d = lambda t: None if t is None else t.double().clone().requires_grad_(True)   # noqa: E731
    lv = dict(means3D=d(sc.means3D), opacity=d(sc.opacity), shs=None if colors is not None else d(sc.shs),
              colors=d(colors), scales=None if cov is not None else d(sc.scales),
              rotations=None if cov is not None else d(sc.rotations), cov=d(cov), sem=d(semantics))
    kw = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in okw.items()}
    out = ts.rasterize(lv["means3D"], lv["opacity"], shs=lv["shs"], colors_precomp=lv["colors"],
                       scales=lv["scales"], rotations=lv["rotations"], cov3D_precomp=lv["cov"],
                       semantics=lv["sem"], **kw)
    loss = (out["color"] * gc.double()).sum() + (out["depth"] * gd.double()).sum() + (out["alpha"] * ga.double()).sum()
    if semantics is not None:
        loss = loss + (out["semantic"] * gs.double()).sum()
    m2 = out["pre"]["means2D"]          # pixel coordinates; the op reports d/d(NDC): x 0.5 W, 0.5 H
    culled_all = not loss.requires_grad   # every Gaussian culled: the images are constants, every gradient is
    if not culled_all:                    # exactly zero -- the sweep draws such frames
        m2.retain_grad()
        loss.backward()
    zeros = lambda t: np.zeros(tuple(t.shape))                                         # noqa: E731
    g = lambda k: None if lv[k] is None else (zeros(lv[k]) if lv[k].grad is None else lv[k].grad.numpy())   # noqa: E731
    half = np.array([0.5 * int(okw["image_width"]), 0.5 * int(okw["image_height"])])
    m2g = zeros(m2) if culled_all else m2.grad.numpy()
    return dict(dL_dmeans2D_xy=m2g[:, :2] * half[None, :],
                dL_dmeans3D=g("means3D"), dL_dopacity=g("opacity"), dL_dsh=g("shs"), dL_dcolors=g("colors"),
                dL_dscales=g("scales"), dL_drotations=g("rotations"), dL_dcov3D=g("cov"), dL_dsemantic=g("sem"))


def reference_composition_fixture():
    """tests/golden/ref_compose.npz (make_golden.py part_a_compose): raw parameters of a background model
    and two actors, their poses / frames / flip masks, and what the reference's OWN getters
    (StreetGaussianModel.get_xyz / get_rotation / get_scaling / get_opacity / get_features) returned
    for them.  -> (models [ModelParams], poses [None | ActorPose], expected dict)."""
    import torch
    from gaussianrpg_amd.composed import ActorPose, ModelParams
    z = np.load(os.path.join(GOLDEN, "ref_compose.npz"))
    t = lambda k: torch.tensor(z[k])          # noqa: E731
    models, poses = [], []
    for name in ("background", "obj_003", "obj_011"):
        flip = t(name + ".flip") if name != "background" else None
        models.append(ModelParams(*(t("%s.%s" % (name, f)) for f in ModelParams._fields[:6]), flip))
        if name == "background":
            poses.append(None)
        else:
            start, end, scale = (float(v) for v in z[name + ".frames"])
            # gaussian_model_actor.py:74-75
            poses.append(ActorPose(t(name + ".obj_rot"), t(name + ".obj_trans"),
                                   scale * (float(z["frame"]) - start) / (end - start)))
    want = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    return models, poses, want


# ---- the Python -> _C boundary, traced (tests/golden/ref_binding_trace.json) -------------------
class BindingRecorder:
    """A stand-in for the compiled `_C` module that records how a Python wrapper calls it: for every
    call the entry point's name and, per positional argument, WHICH caller-side object it is (tensors are
    recognised by storage address; the empty placeholders by their shape) or its plain value.  Its
    return values are labelled tensors too, so that what a wrapper hands on to the next call, back to the
    caller, or to autograd can be followed.  Used twice: in tests/golden/make_golden.py under the
    reference's own diff_gaussian_rasterization/__init__.py (-> the fixture) and in
    tests/test_binding_trace.py under gaussianrpg_amd/rasterizer.py."""

    def __init__(self):
        import torch
        self.torch = torch
        self.calls = []
        self.names = {}

    def tensor(self, label, *shape, fill=None):
        t = self.torch.full(shape, float(len(self.names) + 1) if fill is None else fill)
        self.names[(t.data_ptr(), tuple(t.shape))] = label
        return t

    def describe(self, a):
        torch = self.torch
        if isinstance(a, torch.Tensor):
            if a.numel() == 0:
                return "empty%s" % (list(a.shape),)
            return self.names.get((a.data_ptr(), tuple(a.shape)), "unknown%s" % (list(a.shape),))
        if isinstance(a, bool):
            return "bool:%s" % a
        if isinstance(a, int):
            return "int:%d" % a
        if isinstance(a, float):
            return "float:%r" % a
        return type(a).__name__

    def _record(self, name, args):
        self.calls.append([name, [self.describe(a) for a in args]])

    # the entry points of rasterize_points.h / ext.cpp
    def rasterize_gaussians(self, *args):
        self._record("rasterize_gaussians", args)
        P = args[1].shape[0]
        H, W = args[13], args[14]
        S = args[3].shape[1]
        return (4321, self.tensor("out.color", 3, H, W), self.tensor("out.depth", 1, H, W),
                self.tensor("out.alpha", 1, H, W), self.tensor("out.semantic", S, H, W),
                self.tensor("out.radii", P), self.tensor("out.geomBuffer", 16), self.tensor("out.binningBuffer", 16),
                self.tensor("out.imgBuffer", 16))

    def rasterize_gaussians_backward(self, *args):
        self._record("rasterize_gaussians_backward", args)
        P = args[1].shape[0]
        M = max(args[16].shape[1] if args[16].dim() == 3 else 0, 1)
        S = args[24].shape[1]
        shapes = [("means2D", (P, 3)), ("colors_precomp", (P, 3)), ("opacities", (P, 1)), ("means3D", (P, 3)),
                  ("cov3Ds_precomp", (P, 6)), ("sh", (P, M, 3)), ("scales", (P, 3)), ("rotations", (P, 4)),
                  ("semantics", (P, S))]
        return tuple(self.tensor("grad." + n, *shp) for n, shp in shapes)

    def mark_visible(self, *args):
        self._record("mark_visible", args)
        return self.tensor("out.visible", args[0].shape[0])

    def rasterize_gaussians_filter(self, *args):
        self._record("rasterize_gaussians_filter", args)
        P = args[0].shape[0]
        return self.tensor("out.filter_radii", P), self.tensor("out.filter_means2D", P, 2)


def binding_trace(wrapper_module, rec, backward_alias=None):
    """Drive a wrapper module (the reference's __init__.py or gaussianrpg_amd/rasterizer.py, its `_C`
    already replaced by `rec`) through the calls the reference's callers make; returns the trace."""
    import torch
    P, H, W, S = 5, 4, 6, 2
    out = {}
    for scenario in ("sh_scales_rotations", "colors_cov3d"):
        rec.calls.clear()
        leaf = lambda label, *shape: rec.tensor("in." + label, *shape).requires_grad_(True)   # noqa: E731
        rs = wrapper_module.GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=0.5, tanfovy=0.25, bg=rec.tensor("rs.bg", 3),
            scale_modifier=0.75, viewmatrix=rec.tensor("rs.viewmatrix", 4, 4),
            projmatrix=rec.tensor("rs.projmatrix", 4, 4), sh_degree=1, campos=rec.tensor("rs.campos", 3),
            prefiltered=False, debug=False)
        ins = dict(means3D=leaf("means3D", P, 3), means2D=leaf("means2D", P, 3), opacities=leaf("opacities", P, 1),
                   semantics=leaf("semantics", P, S))
        if scenario == "sh_scales_rotations":
            ins.update(shs=leaf("sh", P, 4, 3), scales=leaf("scales", P, 3), rotations=leaf("rotations", P, 4))
        else:
            ins.update(colors_precomp=leaf("colors_precomp", P, 3), cov3D_precomp=leaf("cov3Ds_precomp", P, 6))
        rast = wrapper_module.GaussianRasterizer(rs)
        outs = rast(**ins)
        returned = [rec.describe(o) for o in outs]
        gouts = [rec.tensor("gin." + n, *o.shape) for n, o in zip(("color", "radii", "depth", "alpha", "semantic"), outs)]
        diff = [(o, g) for o, g in zip(outs, gouts) if o.requires_grad]
        torch.autograd.backward([o for o, _ in diff], [g for _, g in diff])
        grads = {k: (rec.describe(v.grad) if v.grad is not None else None) for k, v in ins.items()}
        vis = rast.markVisible(rec.tensor("in.positions", P, 3))
        filt = rast.visible_filter(rec.tensor("in.filter_means3D", P, 3), scales=rec.tensor("in.filter_scales", P, 3),
                                   rotations=rec.tensor("in.filter_rotations", P, 4))
        calls = [[(backward_alias or {}).get(n, n), a] for n, a in rec.calls]
        out[scenario] = dict(calls=calls, returned=returned, input_grads=grads,
                             mark_visible_returns=rec.describe(vis), filter_returns=[rec.describe(t) for t in filt])
    return out


def power_never_positive(conic):
    """numpy restatement (fp32 like the device) of blend_math.h splat_power_never_positive: True where the computed
    power2 provably cannot be positive for any pixel offset.  conic: [n, 3] float32 (a, b, c)."""
    f = np.float32
    log2e = f(1.4426950408889634)
    A = (f(-0.5) * log2e) * conic[:, 0]; B = (-log2e) * conic[:, 1]; C = (f(-0.5) * log2e) * conic[:, 2]
    a, c = -A, -C
    k = np.maximum(f(7.62939453125e-06) * (a + c + np.abs(B)), f(1e-12))
    with np.errstate(invalid="ignore"):   # inf - inf of non-finite conics: NaN compares false, like on the device
        return (a > k) & (c > k) & ((a - k) * (c - k) > f(0.25) * B * B)


def device_power2(conic, dx, dy):
    """The device's power2 = fma(dy, fma(C, dy, fl(B dx)), fl(fl(A dx) dx)) on fp32 arrays (blend_math.h pair_power;
    the fused steps through float64, which holds a product of two floats exactly)."""
    f = np.float32
    log2e = f(1.4426950408889634)
    A = (f(-0.5) * log2e) * conic[:, 0]; B = (-log2e) * conic[:, 1]; C = (f(-0.5) * log2e) * conic[:, 2]
    fma = lambda a, b, c: (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)  # noqa: E731
    return fma(dy, fma(C, dy, B * dx), (A * dx) * dx)

