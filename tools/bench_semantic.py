"""Semantic path (S > 0) timing: the reference's smoke case (script/test_gaussian_rasterization.py:73-87:
P = 10 000 random Gaussians, 1242x375, S = 15) and the bench frame (P = 2 M, 1920x1280) with S = 15.
Prints one JSON line per case: forward ms with S = 0 and S = 15 (device time, event pairs)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianrpg_amd import harness as hz
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

dev = torch.device("cuda:0")


def timed(rast, sc, sem, n=20, warm=3):
    ms = []
    for i in range(n + warm):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = rast(means3D=sc.means3D, means2D=None, opacities=sc.opacity, shs=sc.shs, scales=sc.scales,
                   rotations=sc.rotations, semantics=sem)
        e1.record()
        torch.cuda.synchronize()
        if i >= warm:
            ms.append(e0.elapsed_time(e1))
    ms.sort()
    return ms[len(ms) // 2], out


for name, sc, cam in (("smoke P=10000 @1242x375", hz.smoke_scene(10000, seed=0), hz.smoke_camera(1242, 375, device=dev)),
                      ("bench P=2000000 @1920x1280", hz.street_scene(2_000_000, seed=2), hz.trajectory_camera(0, device=dev))):
    sc = sc.to(dev)
    P = sc.means3D.shape[0]
    rast = GaussianRasterizer(GaussianRasterizationSettings(**hz.settings_kwargs(cam, sc.sh_degree)))
    sem = torch.rand(P, 15, device=dev)
    t0, _ = timed(rast, sc, None)
    t15, out = timed(rast, sc, sem)
    from gaussianrpg_amd.rasterizer import _C
    _C.set_stage_timing(1)
    for _ in range(5):
        rast(means3D=sc.means3D, means2D=None, opacities=sc.opacity, shs=sc.shs, scales=sc.scales,
             rotations=sc.rotations, semantics=sem)
    torch.cuda.synchronize()
    st = _C.stage_timing()
    _C.set_stage_timing(0)
    ms, calls = st
    print(json.dumps({"case": name, "S0_forward_ms": t0, "S15_forward_ms": t15, "ratio": t15 / t0,
                      "stage_ms_S15_per_frame": [float(x) / max(calls, 1) for x in ms],
                      "stages": "preprocess, depth sort, coarse scan, coarse emit, coarse partition, counts+fill, render, semantic render"}))
