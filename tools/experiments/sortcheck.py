import sys, os
sys.path.insert(0, os.getcwd())
import torch
from gaussianrpg_amd import harness as hz, trajectory as tj
from gaussianrpg_amd.rasterizer import _C
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
dev = torch.device("cuda:0")
sc = hz.street_scene(2_000_000, seed=2, sh_degree=1).to(dev)
tape = tj.make_tape(200)
for k in list(range(0, 200, 20)) + [0, 1, 2]:
    cam = tj.camera_from_tape(tape[k], W=1920, H=1280, device=dev)
    r = GaussianRasterizer(GaussianRasterizationSettings(**hz.settings_kwargs(cam, 1, bg=torch.zeros(3, device=dev))))
    for _ in range(2):
        r(means3D=sc.means3D, means2D=None, opacities=sc.opacity, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    torch.cuda.synchronize()
    _C.set_stage_timing(1)
    for _ in range(5):
        out = r(means3D=sc.means3D, means2D=None, opacities=sc.opacity, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    torch.cuda.synchronize()
    ms, n = _C.stage_timing()
    _C.set_stage_timing(0)
    print(k, "V", int((out[1] > 0).sum()), "stages", [round(x / n, 4) for x in ms])
