"""One layered frame in a loop, for rocprofv3 --kernel-trace --stats: which kernel of the layered frame costs what.
usage: python tools/prof_layers.py actors|legacy|none [frames]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianrpg_amd import harness as hz
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer


def actor_scene(dev):
    sc, obj = hz.actor_scene()
    return sc.to(dev), obj.to(dev)


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
