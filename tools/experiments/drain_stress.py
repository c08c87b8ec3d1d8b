"""Stress of the drained host frame's hand-over protocol (staging stores -> vmcnt(0) -> arrival counter -> drain waves'
agent-scope reads -> write-through host stores): many frames of random shapes, scene sizes and stream counts, the host
bytes compared with the device-destination bytes of the same call.  A lost or early arrival shows up as stale bytes
(the host frame is poisoned before every call) or as the render's timeout error.  usage: drain_stress.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gaussianrpg_amd import harness as hz
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(123)
scene = hz.street_scene(400_000, seed=7).to(dev)
shapes = [(1920, 1280), (1600, 1066), (1280, 720), (960, 540), (640, 360), (1936, 400), (400, 300), (2560, 1440), (336, 1000)]
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
t0, frames, bad = time.time(), 0, 0
with torch.no_grad():
    while time.time() - t0 < budget:
        W, H = shapes[int(rng.integers(len(shapes)))]
        n = int(rng.integers(2_000, 400_000))
        sel = slice(0, n)
        kw = dict(shs=scene.shs[sel], scales=scene.scales[sel], rotations=scene.rotations[sel])
        ns = int(rng.integers(1, 4))
        cams = [hz.trajectory_camera(int(rng.integers(0, 50)), W=W, H=H, device=dev) for _ in range(ns)]
        rasts = [GaussianRasterizer(GaussianRasterizationSettings(**hz.settings_kwargs(c, 1))) for c in cams]
        hosts = [torch.full((H, W, 3), 0xA5, dtype=torch.uint8).pin_memory() for _ in range(ns)]
        refs = [r.forward_frame(scene.means3D[sel], scene.opacity[sel], **kw)["rgb8"] for r in rasts]
        torch.cuda.synchronize()
        for rep in range(3):
            for h in hosts:
                h.fill_(0xA5)
            for st in streams:
                st.wait_stream(torch.cuda.current_stream())
            for k in range(ns):
                with torch.cuda.stream(streams[k]):
                    rasts[k].forward_frame(scene.means3D[sel], scene.opacity[sel], out=hosts[k], **kw)
            torch.cuda.synchronize()
            for k in range(ns):
                frames += 1
                if not torch.equal(hosts[k], refs[k].cpu()):
                    bad += 1
                    d = (hosts[k] != refs[k].cpu())
                    print("MISMATCH %dx%d P=%d streams=%d rep=%d: %d bytes differ, first at %s" % (
                        W, H, n, ns, rep, int(d.sum()), d.nonzero()[0].tolist()), flush=True)
print("frames %d, mismatching %d, %.0f s" % (frames, bad, time.time() - t0))
sys.exit(1 if bad else 0)
