"""Where the HOST time of one simulator frame goes (bench.py simulator_frame, the one_call_rgb_host form): per frame,
time.perf_counter() around the pose objects, the ray matrix, composed._pack, the binding call (returns when the frame
is enqueued) and the final synchronize.  The GPU is idle until the binding's first launch: everything in front of it is
on the frame's critical path in a closed loop."""
import json, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gaussianrpg_amd import composed as C, harness as hz
from gaussianrpg_amd.sky import SkyCubeMap, ray_matrix_host
from diff_gaussian_rasterization import GaussianRasterizationSettings

dev = torch.device("cuda:0")
W, H = bench.W, bench.H
g = torch.Generator().manual_seed(2)
NB, NA, PA, F = 1_900_000, 10, 10_000, 5
sc = hz.street_scene(NB, seed=2)
op = sc.opacity.clamp(1e-4, 1 - 1e-4)
models = [C.ModelParams(sc.means3D, torch.log(sc.scales), sc.rotations * 1.7, torch.log(op / (1 - op)),
                        sc.shs[:, :1].contiguous(), sc.shs[:, 1:].contiguous())]
for _ in range(NA):
    models.append(C.ModelParams((torch.rand(PA, 3, generator=g) - 0.5) * torch.tensor([4.5, 1.6, 2.0]),
                                math.log(0.05) + 0.5 * torch.randn(PA, 3, generator=g), torch.randn(PA, 4, generator=g),
                                1.0 + 2.0 * torch.randn(PA, 1, generator=g), 0.5 * torch.randn(PA, F, 3, generator=g),
                                0.15 * torch.randn(PA, 3, 3, generator=g)))
models = [C.ModelParams(*(t.to(dev) for t in m[:6])) for m in models]
sky = SkyCubeMap(resolution=1024).to(dev)
N = 60
cams = [hz.trajectory_camera(f, device=dev) for f in range(N)]
crs = [C.ComposedRasterizer(GaussianRasterizationSettings(**hz.settings_kwargs(c, 1))) for c in cams]
Ks = [torch.tensor([[W / (2.0 * c.tanfovx), 0.0, W / 2.0], [0.0, H / (2.0 * c.tanfovy), H / 2.0], [0.0, 0.0, 1.0]]) for c in cams]
w2cs = [c.viewmatrix.t().contiguous().cpu() for c in cams]
host = torch.empty((H, W, 3), dtype=torch.uint8).pin_memory()
acc = {k: [] for k in ("poses", "ray_matrix", "pack", "binding_call", "synchronize", "frame")}
real_pack = C._pack
with torch.no_grad():
    for f in range(N):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        poses = [None] + [C.ActorPose([math.cos(0.05 * k + 0.002 * f), 0.0, math.sin(0.05 * k + 0.002 * f), 0.0],
                                      [-12.0 + 2.5 * k, 0.8, 10.0 + 8.0 * k + 0.5 * f], 0.1 + 0.004 * f) for k in range(NA)]
        t1 = time.perf_counter()
        rm = ray_matrix_host(Ks[f], w2cs[f])
        t2 = time.perf_counter()
        packed = real_pack(models, poses)
        t3 = time.perf_counter()
        C._pack = lambda m, p: packed          # the call below reuses it: its own time is the binding's
        crs[f].forward_frame(models, poses, sky_cube=sky.sky_cube_map, ray_matrix=rm, layers=False, planes=False, out=host)
        C._pack = real_pack
        t4 = time.perf_counter()
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        if f >= 10:
            for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0)):
                acc[k].append(1e6 * v)
med = {k: sorted(v)[len(v) // 2] for k, v in acc.items()}
print(json.dumps({"median_us": med, "what": __doc__.split("\n")[0]}))
