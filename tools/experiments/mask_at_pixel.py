"""For one forward draw of tests/test_gpu_sweep.py and one pixel: every entry of the pixel's tile list that
the oracle BLENDS at this pixel, with the sub-tile mask bit the HIP point list carries for the pixel's
quarter.  A blended entry whose bit is clear is a splat the render never loads: a mask bug.
    python tools/experiments/mask_at_pixel.py <seed> <y> <x>      (needs the GPU)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle                                         # noqa: E402
from gaussianrpg_amd import harness as hz             # noqa: E402
from helpers import oracle_kwargs                     # noqa: E402
from test_gpu_sweep import draw                       # noqa: E402

seed, y, x = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
d = draw(seed, 60000, 420)
sc, cam = d["sc"], d["cam"]
o = oracle.forward(sc.means3D, sc.opacity, shs=sc.shs, scales=sc.scales, rotations=sc.rotations,
                   **oracle_kwargs(cam, sc.sh_degree, bg=d["bg"], scale_modifier=d["scale_modifier"]))
from diff_gaussian_rasterization import GaussianRasterizationSettings   # noqa: E402
from gaussianrpg_amd.rasterizer import _C                               # noqa: E402
dev = torch.device("cuda:0")
camd = hz.CameraTensors(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, cam.viewmatrix.to(dev),
                        cam.projmatrix.to(dev), cam.campos.to(dev))
rs = GaussianRasterizationSettings(**hz.settings_kwargs(camd, sc.sh_degree, bg=d["bg"].to(dev),
                                                        scale_modifier=d["scale_modifier"]))
dd = sc.to(dev)
e = torch.Tensor([])
P = dd.means3D.shape[0]
out = _C.rasterize_gaussians(rs.bg, dd.means3D, e, torch.zeros(P, 0, device=dev), dd.opacity, dd.scales, dd.rotations,
                             rs.scale_modifier, e, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy,
                             rs.image_height, rs.image_width, dd.shs, rs.sh_degree, rs.campos, False, False)
torch.cuda.synchronize()
R, binning = out[0], out[7]
raw = binning.cpu().numpy().view(np.uint32)[64:64 + R]          # the point list: first array behind the 256-byte header
H, W = cam.image_height, cam.image_width
gx = (W + 15) // 16
rg = np.asarray(o["ranges"]).astype(np.int64).reshape(-1, 2)
pl = np.asarray(o["point_list"]).astype(np.int64)
t = (y // 16) * gx + x // 16
b, en = rg[t]
assert np.array_equal(raw[b:en] & 0x0FFFFFFF, pl[b:en]), "lists differ"
q = (y % 16) // 4
m2, co = np.asarray(o["means2D"], np.float32), np.asarray(o["conic_opacity"], np.float32)
f32 = np.float32
T = f32(1.0)
bad = 0
for k in range(b, en):
    i = pl[k]
    dx, dy = f32(m2[i, 0] - f32(x)), f32(m2[i, 1] - f32(y))
    power = f32(f32(-0.5) * (co[i, 0] * dx * dx + co[i, 2] * dy * dy) - co[i, 1] * dx * dy)
    if power > 0:
        continue
    alpha = min(f32(0.99), f32(co[i, 3] * np.exp(power)))
    if alpha < f32(1 / 255):
        continue
    tt = f32(T * f32(1 - alpha))
    if tt < 1e-4:
        break
    bit = (int(raw[k]) >> (28 + q)) & 1
    if not bit:
        bad += 1
        print("MASK CLEAR for a blended splat: k", k - b, "id", i, "alpha %.5e T %.4e" % (alpha, T), "mask bits %x" % (int(raw[k]) >> 28),
              "mean2D", m2[i], "conic", co[i, :3], "opacity", co[i, 3], "radius", o["radii"][i], "scales", sc.scales[i].numpy())
    T = tt
print("pixel", (y, x), "tile", t, "quarter", q, "entries", en - b, "blended entries with a clear mask bit:", bad)
