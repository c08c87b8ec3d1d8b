"""Phase timing of the depth sort's scatter passes (experiment build -DGRPG_DS_TRACE).
usage on the GPU box:  LD_PRELOAD=build/variants/libgrpg_rasterizer_dstrace.so python tools/ds_trace.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gaussianrpg_amd import harness as hz
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
dev = torch.device("cuda:0")
sc = hz.street_scene(2_000_000, seed=2, sh_degree=1).to(dev)
lib = ctypes.CDLL(os.path.join(ROOT, "build", "variants", "libgrpg_rasterizer_dstrace.so"))
names = ["start", "loads issued", "sweep done", "barrier", "digit bases", "ranked", "barrier", "staged", "written"]
for k in range(4):
    cam = hz.trajectory_camera(k, device=dev)
    r = GaussianRasterizer(GaussianRasterizationSettings(**hz.settings_kwargs(cam, 1)))
    r(means3D=sc.means3D, means2D=None, opacities=sc.opacity, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (4 * 4 * 10))()
print("rc", lib.grpg_debug_ds_trace(buf))
for p in range(3):
    t0s = [buf[(p * 4 + w) * 10 + 0] for w in range(4)]
    t8s = [buf[(p * 4 + w) * 10 + 8] for w in range(4)]
    base = min(t0s)
    print("pass %d: workgroups 0 / 50 / 120 / 170 start at +%s us, end at +%s us of the first start" % (
        p, " / ".join("%.2f" % ((t - base) / 100.0) for t in t0s), " / ".join("%.2f" % ((t - base) / 100.0) for t in t8s)))
    for w in range(4):
        t = [buf[(p * 4 + w) * 10 + i] for i in range(9)]
        print("pass %d wg-probe %d: " % (p, w) + "  ".join("%s +%.2fus" % (names[i], (t[i] - t[i - 1]) / 100.0) for i in range(1, 9)) + "  | total %.2f us" % ((t[8] - t[0]) / 100.0))
