"""Debug tool (GPU box): dump per-wave cycle counts of the render kernel for bench frame 0.
The trace code only exists in the experiment build (python -c "from gaussianrpg_amd import build;
build.build_variant('trace')"):
usage: LD_PRELOAD=$PWD/build/variants/libgrpg_rasterizer_trace.so GRPG_RENDER_TRACE=gpurun_out/trace.bin \
       python tools/trace_render.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianrpg_amd import harness as hz
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
dev = torch.device("cuda:0")
sc = hz.street_scene(2_000_000, seed=2).to(dev)
cam = hz.trajectory_camera(0, device=dev)
r = GaussianRasterizer(GaussianRasterizationSettings(**hz.settings_kwargs(cam, 1)))
for _ in range(2):
    r(means3D=sc.means3D, means2D=None, opacities=sc.opacity, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
torch.cuda.synchronize()
print("trace written to", os.environ.get("GRPG_RENDER_TRACE"))
