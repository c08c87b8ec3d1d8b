"""How far apart do two fits land that differ only by rounding / atomic order?  (Sizing of the
PSNR-parity bar of tests/test_gpu_backward.py::test_fit_psnr_parity_10k.)"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from oracle import torch_splat as ts
from gaussianrpg_amd import harness as hz
from helpers import oracle_kwargs
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
W, H, P, STEPS = 160, 120, 10000, 200
dev = torch.device("cuda:0")
cam = hz.trajectory_camera(0, W=W, H=H); kw = oracle_kwargs(cam, 0)
target_sc = hz.toy_scene(30000, seed=60, sh_degree=0, scale=0.04, spread=2.0)
start = hz.toy_scene(P, seed=61, sh_degree=0, scale=0.06, spread=2.0)
oracle.use_openmp(True)
target = torch.from_numpy(oracle.forward(target_sc.means3D, target_sc.opacity, shs=target_sc.shs, scales=target_sc.scales, rotations=target_sc.rotations, **kw)["color"].copy())
class OS(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m, o, s, sc, r):
        out = oracle.forward(m.detach(), o.detach(), shs=s.detach(), scales=sc.detach(), rotations=r, **kw); ctx.o = out
        return torch.from_numpy(out["color"].copy())
    @staticmethod
    def backward(ctx, g):
        z = np.zeros((H, W), np.float32); gg = oracle.backward(ctx.o, g.contiguous().numpy(), z, z)
        t = lambda k: torch.from_numpy(gg[k])
        return t("dL_dmeans3D"), t("dL_dopacity"), t("dL_dsh"), t("dL_dscales"), None
def fit(render, dev_, lr):
    p = {k: getattr(start, k).to(dev_).clone().requires_grad_(True) for k in ("means3D", "opacity", "shs", "scales")}
    rot = start.rotations.to(dev_); opt = torch.optim.Adam(p.values(), lr=lr); tgt = target.to(dev_)
    out = []
    for i in range(STEPS + 1):
        opt.zero_grad()
        img = render(p["means3D"], p["opacity"].clamp(0.01, 0.99), p["shs"], p["scales"].clamp(0.01, 2.0), rot)
        if i in (50, 100, 200): out.append(round(ts.psnr(img.detach().clamp(0, 1).cpu(), tgt.clamp(0, 1).cpu()), 4))
        if i == STEPS: break
        (img - tgt).abs().mean().backward(); opt.step()
    return out
camd = hz.trajectory_camera(0, W=W, H=H, device=dev)
rast = GaussianRasterizer(GaussianRasterizationSettings(**hz.settings_kwargs(camd, 0)))
hip = lambda m, o, s, sc_, r: rast(means3D=m, means2D=None, opacities=o, shs=s, scales=sc_, rotations=r)[0]
for lr in (0.002, 0.005):
    a = fit(hip, dev, lr); b = fit(hip, dev, lr); c = fit(lambda *x: OS.apply(*x), torch.device("cpu"), lr)
    print("lr", lr, "hip_a", a, "hip_b", b, "oracle", c, flush=True)
