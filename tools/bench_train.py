"""Config 5 of BASELINE.json: forward + backward of the op on the scene-149-like synthetic scene
(P = 1 M, 1920x1280), train-mode argument pattern (means2D = zeros[P,3].requires_grad,
lib/models/street_gaussian_renderer.py:157-162) and a loss that makes all four output gradients
non-zero (L1 to a fixed target + depth/acc term, mirrors train.py:116-118,164-176).
Prints one JSON line: forward ms, backward ms (torch.cuda.synchronize-bracketed, per iteration)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianrpg_amd import harness as hz
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

ap = argparse.ArgumentParser()
ap.add_argument("--gaussians", type=int, default=1_000_000)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda:0")
sc = hz.street_scene(args.gaussians, seed=149).to(dev)
leaves = [t.clone().requires_grad_(True) for t in (sc.means3D, sc.opacity, sc.shs, sc.scales, sc.rotations)]
target = torch.rand(3, hz.WAYMO_H, hz.WAYMO_W, device=dev)
fw, bw, ob = [], [], []
for it in range(args.warmup + args.steps):
    cam = hz.trajectory_camera(it % 200, device=dev)
    rast = GaussianRasterizer(GaussianRasterizationSettings(**hz.settings_kwargs(cam, 1)))
    means2D = torch.zeros(args.gaussians, 3, device=dev, requires_grad=True)
    for t in leaves:
        t.grad = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    color, radii, depth, alpha, sem = rast(means3D=leaves[0], means2D=means2D, opacities=leaves[1],
                                           shs=leaves[2], scales=leaves[3], rotations=leaves[4])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss = (color - target).abs().mean() + 0.1 * (depth / (alpha + 1e-10)).mean() + 0.05 * alpha.mean()
    outs = (color, depth, alpha)
    gouts = torch.autograd.grad(loss, outs, retain_graph=True)     # the loss' own backward, untimed
    torch.cuda.synchronize(); t2 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.autograd.backward(outs, gouts)                           # the op's backward alone
    e1.record()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    if it >= args.warmup:
        fw.append(t1 - t0); bw.append(t3 - t2); ob.append(e0.elapsed_time(e1))
fw.sort(); bw.sort(); ob.sort()
print(json.dumps({"config": "configs[4]: train fwd+bwd, scene-149-like P=%d @1920x1280" % args.gaussians,
                  "forward_ms_median": 1e3 * fw[len(fw) // 2], "backward_ms_median": 1e3 * bw[len(bw) // 2],
                  "op_backward_device_ms_median": ob[len(ob) // 2],
                  "backward_includes": "_C.rasterize_gaussians_backward alone (wall, and device time between two events)",
                  "steps": args.steps}))
