"""How many frames per second can ONE host thread enqueue?  A scene so small that the GPU is idle most
of the time (P = 2000 at 320x200) through the same Python entry points the bench loop uses."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gaussianrpg_amd import harness as hz, trajectory as tj
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
dev = torch.device("cuda:0")
sc = hz.toy_scene(2000, seed=1).to(dev)
cam = hz.trajectory_camera(0, W=320, H=200, device=dev)
r = GaussianRasterizer(GaussianRasterizationSettings(**hz.settings_kwargs(cam, 1, bg=torch.zeros(3, device=dev))))
kw = dict(means3D=sc.means3D, opacities=sc.opacity, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
streams = [torch.cuda.Stream() for _ in range(3)]
out = torch.empty(3, 200, 320, dtype=torch.uint8, device=dev)
for name in ("forward", "forward+pack", "deferred+pack"):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        N = 600
        for i in range(N):
            with torch.cuda.stream(streams[i % 3]):
                if name == "forward":
                    r(means2D=None, **kw)
                elif name == "forward+pack":
                    tj.pack_u8(r(means2D=None, **kw)[0], out=out)
                else:
                    t = r.forward_deferred(**kw); tj.pack_u8(t[1], out=out)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-14s %.1f frames/s  (%.1f us per frame on the host)" % (name, N / dt, dt / N * 1e6))
