"""One layered frame in a loop, for rocprofv3 --kernel-trace --stats: which kernel of the layered frame costs what.
usage: python tools/prof_layers.py actors|legacy|none [frames]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianrpg_amd import harness as hz
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer


def actor_scene(dev, NB=1_900_000, NA=10, PA=10_000, seed=2):
    """tools/bench_layers.py's second case: background + ten car-sized boxes of small Gaussians on the road."""
    g = torch.Generator().manual_seed(seed)
    parts = [hz.street_scene(NB, seed=seed)]
    for k in range(NA):
        a = 0.05 * k
        loc = (torch.rand(PA, 3, generator=g) - 0.5) * torch.tensor([4.5, 1.6, 2.0])
        ca, sa = math.cos(a), math.sin(a)
        rot = torch.tensor([[ca, 0.0, sa], [0.0, 1.0, 0.0], [-sa, 0.0, ca]])
        xyz = loc @ rot.T + torch.tensor([-12.0 + 2.5 * k, 0.8, 10.0 + 8.0 * k])
        q = torch.nn.functional.normalize(torch.randn(PA, 4, generator=g), dim=1)
        parts.append(hz.Scene(xyz, torch.sigmoid(1.0 + 2.0 * torch.randn(PA, 1, generator=g)),
                              torch.exp(math.log(0.05) + 0.5 * torch.randn(PA, 3, generator=g)), q,
                              torch.cat((0.5 * torch.randn(PA, 1, 3, generator=g), 0.15 * torch.randn(PA, 3, 3, generator=g)), 1), 1))
    cat = lambda k: torch.cat([getattr(p, k) for p in parts]).contiguous().to(dev)   # noqa: E731
    sc = hz.Scene(cat("means3D"), cat("opacity"), cat("scales"), cat("rotations"), cat("shs"), 1)
    obj = torch.zeros(sc.means3D.shape[0], dtype=torch.bool, device=dev)
    obj[NB:] = True
    return sc, obj


def legacy_scene(dev):
    sc = hz.street_scene(2_000_000, seed=2).to(dev)
    obj = torch.zeros(sc.means3D.shape[0], dtype=torch.bool, device=dev)
    for k in range(10):
        c = torch.tensor([(-1) ** k * 2.0, 1.0, 8.0 + 6.0 * k], device=dev)
        obj[torch.topk(((sc.means3D - c) ** 2).sum(1), 10000, largest=False).indices] = True
    return sc, obj


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "actors"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    sc, obj = actor_scene(dev) if which == "actors" else legacy_scene(dev)
    if which == "none":
        obj = torch.zeros_like(obj)
    cam = hz.trajectory_camera(0, device=dev)
    rast = GaussianRasterizer(GaussianRasterizationSettings(**hz.settings_kwargs(cam, 1)))
    with torch.no_grad():
        for _ in range(n):
            rast.forward_layers(sc.means3D, sc.opacity, obj, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
        torch.cuda.synchronize()
