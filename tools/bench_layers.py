"""Layered forward (grpg_forward_layers) on the bench frame with ten actor-sized clusters marked as objects:
device time per frame and the stage breakdown, next to the plain forward of the same scene.  One JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianrpg_amd import harness as hz
from gaussianrpg_amd.rasterizer import _C
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

dev = torch.device("cuda:0")
sc = hz.street_scene(2_000_000, seed=2).to(dev)
P = sc.means3D.shape[0]
obj = torch.zeros(P, dtype=torch.bool, device=dev)
for k in range(10):
    c = torch.tensor([(-1) ** k * 2.0, 1.0, 8.0 + 6.0 * k], device=dev)
    obj[torch.topk(((sc.means3D - c) ** 2).sum(1), 10000, largest=False).indices] = True
cam = hz.trajectory_camera(0, device=dev)
rast = GaussianRasterizer(GaussianRasterizationSettings(**hz.settings_kwargs(cam, 1)))
kw = dict(shs=sc.shs, scales=sc.scales, rotations=sc.rotations)


def timed(fn, n=20, warm=3):
    ms = []
    for i in range(n + warm):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        if i >= warm:
            ms.append(e0.elapsed_time(e1))
    ms.sort()
    return ms[len(ms) // 2]


def stages(fn):
    _C.set_stage_timing(1)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ms, calls = _C.stage_timing()
    _C.set_stage_timing(0)
    return [float(x) / max(calls, 1) for x in ms]


with torch.no_grad():
    plain = lambda: rast(means3D=sc.means3D, means2D=None, opacities=sc.opacity, **kw)      # noqa: E731
    layered = lambda: rast.forward_layers(sc.means3D, sc.opacity, obj, **kw)                # noqa: E731
    none = torch.zeros_like(obj)
    layered_none = lambda: rast.forward_layers(sc.means3D, sc.opacity, none, **kw)          # noqa: E731
    o = layered()
    # share of the 16x16 tiles some object pixel lands in (a lower bound of the tiles whose LIST holds an object entry)
    H, W = o["alpha_object"].shape[-2:]
    a = torch.nn.functional.pad(o["alpha_object"][0], (0, (-W) % 16, 0, (-H) % 16))
    tiles_hit = (a.reshape(a.shape[0] // 16, 16, a.shape[1] // 16, 16).amax(dim=(1, 3)) > 0)
    out = {"case": "bench P=2000000 @1920x1280, %d object Gaussians" % int(obj.sum()),
           "forward_ms": timed(plain), "forward_layers_ms": timed(layered),
           "forward_layers_no_objects_ms": timed(layered_none),
           "tiles_with_object_pixels_share": float(tiles_hit.float().mean()),
           "stage_ms_forward": stages(plain), "stage_ms_layers": stages(layered),
           "stage_ms_layers_no_objects": stages(layered_none),
           "stages": "preprocess, depth sort, coarse scan, coarse emit, coarse partition, counts+fill, render "
                     "(layers: class marks + three-state render), semantic render"}

# Second case: ACTOR-like objects -- ten car-sized boxes (4.5 x 1.6 x 2 m) of 10 k small Gaussians (sigma ~ 5 cm) standing
# on the road ahead, the shape tools/bench_compose.py poses its actor models in ("10 actors of
# 10 k"), appended to a 1.9 M background.  The first case marks the 10 k scene Gaussians NEAREST to ten road points
# as objects, which sweeps up metre-sized ground splats: two thirds of the frame's tiles then hold object entries.
sc2, obj2 = hz.actor_scene()
sc2, obj2 = sc2.to(dev), obj2.to(dev)
kw2 = dict(shs=sc2.shs, scales=sc2.scales, rotations=sc2.rotations)
with torch.no_grad():
    plain2 = lambda: rast(means3D=sc2.means3D, means2D=None, opacities=sc2.opacity, **kw2)   # noqa: E731
    layered2 = lambda: rast.forward_layers(sc2.means3D, sc2.opacity, obj2, **kw2)            # noqa: E731
    o2 = layered2()
    a2 = torch.nn.functional.pad(o2["alpha_object"][0], (0, (-W) % 16, 0, (-H) % 16))
    hit2 = (a2.reshape(a2.shape[0] // 16, 16, a2.shape[1] // 16, 16).amax(dim=(1, 3)) > 0)
    st2 = stages(layered2)
    # the floor of the background layer's work: the plain op on the background model alone (what the second of the
    # reference's three calls renders)
    NBG = 1_900_000
    bgonly = lambda: rast(means3D=sc2.means3D[:NBG], means2D=None, opacities=sc2.opacity[:NBG], shs=sc2.shs[:NBG],   # noqa: E731
                          scales=sc2.scales[:NBG], rotations=sc2.rotations[:NBG])
    st_bg = stages(bgonly)
    st_pl = stages(plain2)
    out.update({"actors_case": "1.9 M background + 10 actor boxes of 10 k Gaussians (P = 2 000 000)",
                "actors_forward_ms": timed(plain2), "actors_forward_layers_ms": timed(layered2),
                "actors_tiles_with_object_pixels_share": float(hit2.float().mean()),
                "actors_render_stage_ms": st2[6], "actors_stage_ms_layers": st2,
                "actors_plain_render_stage_ms": st_pl[6], "actors_background_only_render_stage_ms": st_bg[6],
                "actors_object_pixels_share": float((o2["alpha_object"] > 0).float().mean())})
print(json.dumps(out))
