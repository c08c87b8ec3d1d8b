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
    out = {"case": "bench P=2000000 @1920x1280, %d object Gaussians" % int(obj.sum()),
           "forward_ms": timed(plain), "forward_layers_ms": timed(layered),
           "stage_ms_forward": stages(plain), "stage_ms_layers": stages(layered),
           "stages": "preprocess, depth sort, coarse scan, coarse emit, coarse partition, counts+fill, render "
                     "(layers: class marks + three-state render), semantic render"}
print(json.dumps(out))
