"""Hashes of the planes of bench frames (evaluation and training-mode forward), for bit-for-bit comparisons of
two library builds on one box:  python tools/frame_hash.py  vs  LD_PRELOAD=build/variants/lib..._X.so python tools/frame_hash.py"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussianrpg_amd import harness as hz
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
dev = torch.device("cuda:0")
def h(t):
    return hashlib.sha1(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:12]
for P, frames in ((2_000_000, (0, 57, 140)), (300_000, (3,))):
    sc = hz.street_scene(P, seed=2).to(dev)
    for f in frames:
        cam = hz.trajectory_camera(f, device=dev)
        r = GaussianRasterizer(GaussianRasterizationSettings(**hz.settings_kwargs(cam, 1)))
        out = r(means3D=sc.means3D, means2D=None, opacities=sc.opacity, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
        m3 = sc.means3D.clone().requires_grad_(True)   # training-mode forward: n_contrib and checkpoints are written
        outt = r(means3D=m3, means2D=torch.zeros_like(m3, requires_grad=True), opacities=sc.opacity, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
        print(P, f, "eval", [h(x) for x in out[:4]], "train", [h(x) for x in outt[:4]])
