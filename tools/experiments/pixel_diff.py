"""For one forward draw of tests/test_gpu_sweep.py: the pixels where the HIP image and the oracle's differ
beyond 1e-4, with the oracle's per-splat decisions at that pixel (which accept / terminate test sits how
close to its threshold).     python tools/experiments/pixel_diff.py <seed>      (needs the GPU)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle                                         # noqa: E402
from helpers import oracle_kwargs                     # noqa: E402
from test_gpu_forward import _rasterize               # noqa: E402
from test_gpu_sweep import draw                       # noqa: E402

seed = int(sys.argv[1])
d = draw(seed, 60000, 420)
sc, cam = d["sc"], d["cam"]
P = sc.means3D.shape[0]
sem = torch.rand(P, d["S"], generator=torch.Generator().manual_seed(seed)) if d["S"] else None
o = oracle.forward(sc.means3D, sc.opacity, shs=sc.shs, scales=sc.scales, rotations=sc.rotations, semantics=sem,
                   **oracle_kwargs(cam, sc.sh_degree, bg=d["bg"], scale_modifier=d["scale_modifier"]))
got = _rasterize(torch.device("cuda:0"), sc, cam, bg=d["bg"], semantics=sem, scale_modifier=d["scale_modifier"])
H, W = cam.image_height, cam.image_width
frag = np.asarray(o["fragile"]).reshape(H, W) != 0
gx = (W + 15) // 16
pl = np.asarray(o["point_list"]).astype(np.int64)
rg = np.asarray(o["ranges"]).astype(np.int64).reshape(-1, 2)
m2, co, dep = np.asarray(o["means2D"], np.float32), np.asarray(o["conic_opacity"], np.float32), np.asarray(o["depths"])
f32 = np.float32
for plane in ("color", "depth", "alpha", "semantic"):
    if not o[plane].size:
        continue
    ref, g = np.asarray(o[plane], np.float64), np.asarray(got[plane], np.float64)
    bad = (np.abs(g - ref) > 1e-4 + 1e-4 * np.abs(ref)).any(axis=0) & ~frag
    for y, x in zip(*np.nonzero(bad)):
        t = (y // 16) * gx + x // 16
        b, e = rg[t]
        print(plane, "pixel", (y, x), "ref", ref[:, y, x], "hip", g[:, y, x], "n_contrib oracle/hip",
              o["n_contrib"].reshape(H, W)[y, x], got["n_contrib"].view(np.uint32).reshape(H, W)[y, x], "list", e - b)
        T = f32(1.0)
        for k in range(b, e):
            i = pl[k]
            dx, dy = f32(m2[i, 0] - f32(x)), f32(m2[i, 1] - f32(y))
            power = f32(f32(-0.5) * (co[i, 0] * dx * dx + co[i, 2] * dy * dy) - co[i, 1] * dx * dy)
            alpha = min(f32(0.99), f32(co[i, 3] * np.exp(power)))
            note = ""
            if power > 0:
                note = "power>0"
            elif alpha < f32(1 / 255):
                note = "alpha<1/255 (%.3e of it)" % (alpha * 255)
            else:
                tt = f32(T * f32(1 - alpha))
                note = "T %.4e -> %.4e" % (T, tt)
                if tt < 1e-4:
                    note += " TERMINATES"
                    print("   k", k - b, "id", i, "power %.5e alpha %.6e depth %.3f" % (power, alpha, dep[i]), note)
                    break
                T = tt
            print("   k", k - b, "id", i, "power %.5e alpha %.6e depth %.3f" % (power, alpha, dep[i]), note)
