"""The headline loop on CU-MASKED streams (hipExtStreamCreateWithCUMask): the chip split into partitions, each
stream confined to one, so that one frame's latency-bound binning chain runs BESIDE the other partitions'
throughput-bound kernels instead of queueing behind them.  Same frames, same op, same pack as bench.py's timed region.
usage: cu_mask_streams.py [frames]      prints frames/s per configuration (two rounds, alternating)"""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gaussianrpg_amd import harness as hz, trajectory as tj
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W, H = 1920, 1280
hip = ctypes.CDLL("libamdhip64.so")
NCU = torch.cuda.get_device_properties(0).multi_processor_count
NX = 8                                         # XCDs; KFD deals the mask's bits round-robin over them: bit i -> XCD i % 8


def masked_stream(pred):
    """A stream whose kernels may only use the CUs i with pred(i) (i = bit index of the mask)."""
    words = (NCU + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(NCU):
        if pred(i):
            mask[i // 32] |= (1 << (i % 32))
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask rc=%d" % rc)
    return torch.cuda.ExternalStream(s.value, device=dev)


def plain(n):
    return [torch.cuda.Stream(device=dev) for _ in range(n)]


CONFIGS = {
    "plain3": lambda: plain(3),
    "xcd_halves_x1": lambda: [masked_stream(lambda i, p=p: (i % NX) // 4 == p) for p in range(2)],
    "xcd_halves_x2": lambda: [masked_stream(lambda i, p=p: (i % NX) // 4 == p) for p in (0, 1, 0, 1)],
    "xcd_quarters_x1": lambda: [masked_stream(lambda i, p=p: (i % NX) // 2 == p) for p in range(4)],
    "xcd_quarters_x2": lambda: [masked_stream(lambda i, p=p: (i % NX) // 2 == p) for p in (0, 1, 2, 3, 0, 1, 2, 3)],
    "cu_halves_x1": lambda: [masked_stream(lambda i, p=p: ((i // NX) % 2) == p) for p in range(2)],
    "cu_halves_x2": lambda: [masked_stream(lambda i, p=p: ((i // NX) % 2) == p) for p in (0, 1, 0, 1)],
    "threequarters+quarter": lambda: [masked_stream(lambda i: (i % NX) < 6), masked_stream(lambda i: (i % NX) >= 6),
                                      masked_stream(lambda i: (i % NX) < 6)],
}

sc = hz.street_scene(2_000_000, seed=2, sh_degree=1).to(dev)
bg = torch.zeros(3, device=dev)
tape = tj.make_tape(200)
rasts = [GaussianRasterizer(GaussianRasterizationSettings(**hz.settings_kwargs(
    tj.camera_from_tape(e, W=W, H=H, device=dev), sc.sh_degree, bg=bg))) for e in tape]
inputs = dict(means3D=sc.means3D, means2D=None, opacities=sc.opacity, shs=sc.shs, scales=sc.scales,
              rotations=sc.rotations, cov3D_precomp=None, semantics=None)
local = torch.empty((K, 3, H, W), dtype=torch.uint8, device=dev)


def loop(streams, n):
    for s in range(n):
        with torch.cuda.stream(streams[s % len(streams)]):
            color = rasts[s % 200](**inputs)[0]
            tj.pack_u8(color, out=local[s % K])
    for st in streams:
        torch.cuda.current_stream().wait_stream(st)


out = {}
which = sys.argv[2] if len(sys.argv) > 2 else "plain3"      # ONE configuration per process: every masked stream is a
with torch.no_grad():                                        # hardware queue of its own, and they are never destroyed here
    streams = CONFIGS[which]()
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
    loop(streams, 2 * len(streams)); torch.cuda.synchronize()
    # is the mask in force?  one frame at a time on stream 0 (a partition of the chip must be slower than the chip)
    lat = []
    for i in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loop(streams[:1], 1)
        torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
    out["one_frame_on_stream0_ms"] = round(1e3 * sorted(lat)[len(lat) // 2], 4)
    for n in (K, 200, K, 200, K, 200):
        loop(streams, 5); torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(streams, n)
        torch.cuda.synchronize()
        out.setdefault("fps_%d" % n, []).append(round(n / (time.perf_counter() - t0), 1))
    out["checksum"] = int(local[(200 - 1) % K].to(torch.int64).sum().item())
print(json.dumps({"config": which, "streams": len(streams), **out}))
