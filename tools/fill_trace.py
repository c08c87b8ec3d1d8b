"""Per-workgroup life of hb_fill_kernel (experiment build -DGRPG_FILL_TRACE).
usage on the GPU box:  LD_PRELOAD=build/variants/libgrpg_rasterizer_filltrace.so python tools/fill_trace.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gaussianrpg_amd import harness as hz
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
dev = torch.device("cuda:0")
sc = hz.street_scene(2_000_000, seed=2, sh_degree=1).to(dev)
lib = ctypes.CDLL(os.path.join(ROOT, "build", "variants", "libgrpg_rasterizer_filltrace.so"))
for k in range(4):
    cam = hz.trajectory_camera(k, device=dev)
    r = GaussianRasterizer(GaussianRasterizationSettings(**hz.settings_kwargs(cam, 1)))
    with torch.no_grad():
        r(means3D=sc.means3D, means2D=None, opacities=sc.opacity, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (1024 * 4))()
print("rc", lib.grpg_debug_fill_trace(buf))
a = np.array(buf[:], dtype=np.uint64).reshape(1024, 4)
a = a[a[:, 2] > 0]
t0 = a[:, 0].min()
start = (a[:, 0] - t0) / 100.0          # wall clock ticks of 10 ns
end = (a[:, 1] - t0) / 100.0
life = end - start
print("workgroups with work: %d; kernel span %.1f us; start: min %.1f median %.1f max %.1f us; end: median %.1f p90 %.1f max %.1f us"
      % (len(a), end.max(), start.min(), np.median(start), start.max(), np.median(end), np.percentile(end, 90), end.max()))
print("life: median %.1f p90 %.1f max %.1f us; segments per workgroup: min %d max %d; instances (wave 0's column) per workgroup: median %d max %d"
      % (np.median(life), np.percentile(life, 90), life.max(), a[:, 2].min(), a[:, 2].max(), np.median(a[:, 3]), a[:, 3].max()))
order = np.argsort(-life)
for i in order[:8]:
    print("  wg %4d: start %.1f end %.1f life %.1f us, %d segments, %d instances in column 0" % (i, start[i], end[i], life[i], a[i, 2], a[i, 3]))
h, edges = np.histogram(end, bins=12)
print("end-time histogram:", list(zip(np.round(edges[:-1], 1), h)))

pb = (ctypes.c_ulonglong * (1024 * 8))()
print("rc", lib.grpg_debug_fill_phase(pb))
ph = np.array(pb[:], dtype=np.uint64).reshape(1024, 8)[:len(life)] / 100.0
names = ["issue loads", "wait for records", "quarter_pre + staging", "barrier", "column compaction", "column walk"]
print("wave 0, per workgroup (us, median over workgroups; sum over its segments): " +
      "  ".join("%s %.1f" % (n, np.median(ph[:, k])) for k, n in enumerate(names)) + "  | sum %.1f" % np.median(ph[:, :6].sum(1)))
