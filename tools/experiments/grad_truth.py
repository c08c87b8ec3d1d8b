"""Who is right when the HIP backward and gs_oracle.c disagree beyond the element-wise bar?

For one draw of tests/test_gpu_sweep.py: gradients from (a) gs_oracle.c (fp32 terms as the reference
writes them, fp64 sums), (b) float64 autograd through oracle/torch_splat.py (the "truth" up to
accept decisions on fragile pixels) and, with a GPU, (c) the HIP op.  Prints, per gradient array,
the share of the non-exempt elements beyond rtol 1e-3 / atol 1e-5 max|ref| for every pair.

    python tools/experiments/grad_truth.py 100 [--hip]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle                                       # noqa: E402
from oracle import torch_splat as ts                # noqa: E402
from helpers import gaussians_fed_by_fragile_pixels, oracle_kwargs   # noqa: E402


def load_draw():
    src = open(os.path.join(ROOT, "tests", "test_gpu_sweep.py")).read().split("@pytest.mark.parametrize")[0]
    src = src.replace("from test_gpu_backward import _run\n", "").replace(
        "from test_gpu_forward import _check, _rasterize\n", "")
    ns = {}
    exec(src, ns)
    return ns["draw"]


def miss(a, b, keep):
    a, b = np.asarray(a, np.float64).reshape(b.shape), np.asarray(b, np.float64)
    scale = np.abs(b).max() + 1e-30
    ratio = (np.abs(a - b) / (1e-5 * scale + 1e-3 * np.abs(b))).reshape(b.shape[0], -1)[keep]
    return float((ratio > 1).mean()), float(ratio.max()) if ratio.size else 0.0


def main():
    seed = int(sys.argv[1])
    hip = "--hip" in sys.argv
    d = load_draw()(seed, 12000, 200)
    sc, cam, bg = d["sc"], d["cam"], d["bg"]
    H, W = cam.image_height, cam.image_width
    g = torch.Generator().manual_seed(100 + seed)
    P = sc.means3D.shape[0]
    S = 0
    gc = torch.randn(3, H, W, generator=g)
    gd = 0.1 * torch.randn(1, H, W, generator=g)
    ga = torch.randn(1, H, W, generator=g)
    okw = oracle_kwargs(cam, sc.sh_degree, bg=bg)
    o = oracle.forward(sc.means3D, sc.opacity, shs=sc.shs, scales=sc.scales, rotations=sc.rotations, **okw)
    ref = oracle.backward(o, gc, gd, ga, torch.zeros(0, H, W))
    fed = gaussians_fed_by_fragile_pixels(o)
    keep = ~np.asarray(fed, bool)
    print("seed", seed, d["kind"], "P", P, "WxH", W, H, "R", o["num_rendered"], "exempt %.3f" % fed.mean())

    # float64 truth
    leaves = {k: getattr(sc, k).double().clone().requires_grad_(True)
              for k in ("means3D", "opacity", "shs", "scales", "rotations")}
    kw64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in okw.items()}
    out = ts.rasterize(leaves["means3D"], leaves["opacity"], shs=leaves["shs"], scales=leaves["scales"],
                       rotations=leaves["rotations"], **kw64)
    loss = (out["color"] * gc.double()).sum() + (out["depth"] * gd.double()).sum() + (out["alpha"] * ga.double()).sum()
    loss.backward()
    truth = dict(dL_dmeans3D=leaves["means3D"].grad, dL_dopacity=leaves["opacity"].grad, dL_dsh=leaves["shs"].grad,
                 dL_dscales=leaves["scales"].grad, dL_drotations=leaves["rotations"].grad)
    got = None
    if hip:
        from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
        from gaussianrpg_amd import harness as hz
        dev = torch.device("cuda:0")
        camd = hz.CameraTensors(H, W, cam.tanfovx, cam.tanfovy, cam.viewmatrix.to(dev), cam.projmatrix.to(dev),
                                cam.campos.to(dev))
        rast = GaussianRasterizer(GaussianRasterizationSettings(**hz.settings_kwargs(camd, sc.sh_degree, bg=bg.to(dev))))
        lv = {k: getattr(sc, k).to(dev).clone().requires_grad_(True)
              for k in ("means3D", "opacity", "shs", "scales", "rotations")}
        m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
        color, radii, depth, alpha, _ = rast(means3D=lv["means3D"], means2D=m2, opacities=lv["opacity"], shs=lv["shs"],
                                             scales=lv["scales"], rotations=lv["rotations"])
        ((color * gc.to(dev)).sum() + (depth * gd.to(dev)).sum() + (alpha * ga.to(dev)).sum()).backward()
        got = dict(dL_dmeans3D=lv["means3D"].grad.cpu(), dL_dopacity=lv["opacity"].grad.cpu(), dL_dsh=lv["shs"].grad.cpu(),
                   dL_dscales=lv["scales"].grad.cpu(), dL_drotations=lv["rotations"].grad.cpu())
    for k in truth:
        t = truth[k].numpy()
        line = "%-14s oracle-vs-f64 miss %.4f worst %.1f" % ((k,) + miss(ref[k], t.reshape(ref[k].shape), keep))
        if got is not None:
            line += " | hip-vs-f64 %.4f %.1f | hip-vs-oracle %.4f %.1f" % (
                miss(got[k].numpy(), t.reshape(ref[k].shape), keep) + miss(got[k].numpy(), ref[k], keep))
        print(line)


if __name__ == "__main__":
    main()
