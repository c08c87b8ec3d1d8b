"""Does a host thread per stream close the gap between `forward` (waits for num_rendered in every call) and
`forward_deferred` in the three-stream loop?  The binding releases the GIL while it waits for the count, so another
thread can enqueue its frame meanwhile.  Same scene / frames / pack as bench.py; K frames, one thread alternating over
the streams vs one thread per stream (frame s -> thread s mod n).  (An earlier round measured "no gain" with two
streams; the sync entry point trails the deferred one by 8 % today.)"""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from gaussianrpg_amd import harness as hz, trajectory as tj
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

dev = torch.device("cuda:0")
W, H = bench.W, bench.H
sc = hz.street_scene(bench.P_GAUSS, seed=bench.SCENE_SEED, sh_degree=1).to(dev)
tape = tj.make_tape(bench.NUM_FRAMES)
bg = torch.zeros(3, device=dev)
rasts = [GaussianRasterizer(GaussianRasterizationSettings(**hz.settings_kwargs(
    tj.camera_from_tape(e, W=W, H=H, device=dev), sc.sh_degree, bg=bg))) for e in tape]
inputs = dict(means3D=sc.means3D, means2D=None, opacities=sc.opacity, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
ns = 3
streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
local = torch.empty((K, 3, H, W), dtype=torch.uint8, device=dev)

def frame(s):
    with torch.cuda.stream(streams[s % ns]):
        tj.pack_u8(rasts[s % bench.NUM_FRAMES](**inputs)[0], out=local[s % K])

def one_thread(n):
    for s in range(n):
        frame(s)

def per_stream_threads(n):
    def work(i):
        with torch.no_grad():
            for s in range(i, n, ns):
                frame(s)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(ns)]
    for t in ts: t.start()
    for t in ts: t.join()

def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)

with torch.no_grad():
    one_thread(12); per_stream_threads(12); torch.cuda.synchronize()
    res = {"frames": K, "one_thread_fps": [], "thread_per_stream_fps": []}
    for rep in range(4):
        res["one_thread_fps"].append(round(timed(one_thread, K), 1))
        res["thread_per_stream_fps"].append(round(timed(per_stream_threads, K), 1))
print(json.dumps(res))
