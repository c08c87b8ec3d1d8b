"""Render stage of the TRAINING forward (n_contrib, blend checkpoints) against the evaluation forward of the same
frames (config 5's scene, P = 1 M, 1920x1280): stage timers of the library, one stream."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gaussianrpg_amd import harness as hz
from gaussianrpg_amd.rasterizer import _C
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
sc = hz.street_scene(P, seed=149).to(dev)
names = ["preprocess", "depth_sort", "coarse_scan", "coarse_emit", "coarse_partition", "tile_count_fill", "render", "semantic"]
out = {}
for mode in ("eval", "train", "eval", "train"):
    leaves = [t.clone().requires_grad_(mode == "train") for t in (sc.means3D, sc.opacity, sc.shs, sc.scales, sc.rotations)]
    def frame(i):
        cam = hz.trajectory_camera(i % 200, device=dev)
        rast = GaussianRasterizer(GaussianRasterizationSettings(**hz.settings_kwargs(cam, 1)))
        m2d = torch.zeros(P, 3, device=dev, requires_grad=True) if mode == "train" else None
        with torch.set_grad_enabled(mode == "train"):
            return rast(means3D=leaves[0], means2D=m2d, opacities=leaves[1], shs=leaves[2], scales=leaves[3], rotations=leaves[4])
    for i in range(4):
        frame(i)
    torch.cuda.synchronize()
    _C.set_stage_timing(1)
    for i in range(20):
        frame(i)
    torch.cuda.synchronize()
    ms, calls = _C.stage_timing()
    _C.set_stage_timing(0)
    out.setdefault(mode, []).append({n: round(m / calls, 4) for n, m in zip(names, ms)})
print(json.dumps(out))
