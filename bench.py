#!/usr/bin/env python
"""Headline benchmark of the hot path: frames/s of the rasterizer forward at 1920x1280 on a
~2 M-Gaussian street scene (BASELINE.json `metric`, configs[2]), 1..8 MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one frame: one call of the drop-in operator (GaussianRasterizer -> _C -> C ABI -> HIP)
on synthetic inputs already resident in HBM, followed by the eval-mode clamp + uint8 pack of the
trajectory mode.  With N GPUs the frames of the synthetic 200-pose drive are sharded round-robin
(frame i -> rank i mod N, gaussianrpg_amd/trajectory.py), every rank renders K frames (weak
scaling) and the ONLY collective is the final RCCL gather of the uint8 frames to rank 0, inside
the timed region.  Rank 0 prints one JSON line.

Extra objects in the line:
  roofline      dominant kernel (render_forward_kernel): algorithmic bytes per launch
                (44*R + 8*T + 20*N, SURVEY.md §8(d) / DESIGN.md §6) / its average duration measured
                with HIP events on the op's own stream during the timed region, vs 8 TB/s.
  frame_roofline  whole-frame B_alg / ms_per_step (the figure BASELINE.json asks for).
  stages_ms     per-stage average device time from the same HIP events.
  cpu_baseline  pure-PyTorch CPU splat (oracle/torch_splat.py) timed on the host cores on a
                bounded sample of the same workload, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from gaussianrpg_amd import harness as hz  # noqa: E402
from gaussianrpg_amd import trajectory as tj  # noqa: E402

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s
NUM_FRAMES = 200               # poses of the synthetic drive (BASELINE config 4)
P_GAUSS = 2_000_000            # "Waymo scene 002 full Street-Gaussians (~2M)" stand-in
SCENE_SEED = 2
W, H = hz.WAYMO_W, hz.WAYMO_H


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--gaussians", type=int, default=P_GAUSS, help="override P (debugging only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-timing", action="store_true")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the frame loop alternates over (independent frames; 1 = serial)")
    ap.add_argument("--gather-batch", type=int, default=25,
                    help="N > 1: frames per asynchronous gather to rank 0 (0 = one gather at the end)")
    ap.add_argument("--no-delivery", action="store_true",
                    help="skip the host-delivery (rgb8 over PCIe) side measurement")
    ap.add_argument("--host-threads", type=int, default=1,
                    help="host threads issuing frames (each alternates over streams/host-threads "
                         "streams); the op releases the GIL while it waits for num_rendered")
    ap.add_argument("--cpu-baseline-worker", type=int, default=0, help=argparse.SUPPRESS)
    return ap.parse_args()


def effective_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota
    (os.cpu_count() reports the host's cores even inside a quota-limited container, and an
    over-subscribed OpenMP pool makes small torch ops hundreds of times slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, n)


def _cpu_baseline_worker(threads):
    """Runs in a child process (hard wall-clock limit enforced by the parent): pure-PyTorch CPU
    splat on every 2nd Gaussian of the bench scene at 960x640.  Prints one JSON line."""
    from oracle import torch_splat as ts
    torch.set_num_threads(threads)
    scene = hz.street_scene(P_GAUSS, seed=SCENE_SEED, sh_degree=1)
    sub = hz.Scene(*(t[::2].contiguous() if isinstance(t, torch.Tensor) else t for t in scene))
    cam = hz.trajectory_camera(0, W=960, H=640)
    kw = hz.settings_kwargs(cam, scene.sh_degree)
    kw.pop("prefiltered"), kw.pop("debug")
    with torch.no_grad():
        t0 = time.time()
        r = ts.rasterize(sub.means3D, sub.opacity, shs=sub.shs, scales=sub.scales,
                         rotations=sub.rotations, **kw)
        dt = time.time() - t0
    print(json.dumps({"seconds": dt, "R_sample": int(r["num_rendered"]),
                      "P_sample": int(sub.means3D.shape[0]), "threads": torch.get_num_threads()}))


def cpu_baseline(R_full, limit_s=240):
    """cpu_baseline leg (rank 0, N=1): the oracle's pure-PyTorch CPU splat, timed on the host
    cores on a bounded sample of the same workload, scaled to whole frames of the full workload
    by the ratio of tile instances (the blend's work is proportional to R)."""
    import subprocess
    cores = effective_cores()
    threads = min(cores, 32)
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker",
                            str(threads)], capture_output=True, text=True, timeout=limit_s)
        res = json.loads(p.stdout.strip().splitlines()[-1])
    except Exception as exc:
        return {"value": None, "unit": "frames/s", "cores": threads, "kind": "port",
                "sample": "cpu baseline did not finish within %d s: %r" % (limit_s, exc)}
    dt, R_s = res["seconds"], max(res["R_sample"], 1)
    est = (1.0 / dt) * (R_s / max(R_full, 1))
    return {"value": est, "unit": "frames/s", "cores": res["threads"], "kind": "port",
            "host_cpu_count": os.cpu_count(), "usable_cores": cores,
            "sample": "oracle/torch_splat.py (pure-PyTorch CPU splat), frame 0, every 2nd Gaussian "
                      "(P=%d) at 960x640: %.2f s for R=%d tile instances on %d threads; value = 1/t "
                      "scaled by R_sample/R_full (R_full=%d) to whole 1920x1280 frames" % (
                          res["P_sample"], dt, R_s, res["threads"], R_full),
            "sample_seconds": dt}


def main():
    args = parse()
    if args.cpu_baseline_worker:
        _cpu_baseline_worker(args.cpu_baseline_worker)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    assert torch.cuda.is_available(), "bench.py needs ROCm devices (no CPU fallback exists)"
    torch.set_num_threads(max(1, min(effective_cores() // max(world, 1), 16)))   # host-side scene synthesis
    # GRPG_BENCH_BACKEND=gloo is a single-GPU debugging aid only (all ranks share cuda:0, the
    # final gather is staged through host memory); the real multi-GPU run uses RCCL ("nccl").
    backend = os.environ.get("GRPG_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", init_method="env://", world_size=world,
                                    rank=rank, device_id=dev)
        else:
            dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank)

    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from gaussianrpg_amd.rasterizer import _C

    scene_cpu = hz.street_scene(args.gaussians, seed=SCENE_SEED, sh_degree=1)
    sc = scene_cpu.to(dev)
    P = sc.means3D.shape[0]
    M = sc.shs.shape[1]
    bg = torch.zeros(3, device=dev)
    tape = tj.make_tape(NUM_FRAMES)
    rasterizers = []
    for e in tape:
        cam = tj.camera_from_tape(e, W=W, H=H, device=dev)
        rasterizers.append(GaussianRasterizer(GaussianRasterizationSettings(
            **hz.settings_kwargs(cam, sc.sh_degree, bg=bg))))

    def render_frame(i):
        # argument pattern of render_kernel in eval mode (street_gaussian_renderer.py:224-237)
        color, radii, depth, alpha, sem = rasterizers[i % NUM_FRAMES](
            means3D=sc.means3D, means2D=None, opacities=sc.opacity, shs=sc.shs, scales=sc.scales,
            rotations=sc.rotations, cov3D_precomp=None, semantics=None)
        return color

    K, Wm = args.steps, args.warmup
    frames_of = lambda s: (s * world + rank)          # noqa: E731  round-robin frame ownership
    local = torch.empty((K, 3, H, W), dtype=torch.uint8, device=dev)

    with torch.no_grad():
        for s in range(Wm):
            tj.pack_u8(render_frame(frames_of(s)))
        gather_bufs = None
        if world > 1:
            import torch.distributed as dist
            cdev = dev if backend == "nccl" else torch.device("cpu")
            if rank == 0:
                gather_bufs = [torch.empty(local.shape, dtype=local.dtype, device=cdev) for _ in range(world)]
            # warm the communicator outside the timed region
            tiny = torch.zeros(1, device=cdev)
            dist.all_reduce(tiny)
        torch.cuda.synchronize()

        stage_timing = not args.no_stage_timing
        # timed region: only the two events around the render stage (the dominant kernel, needed for
        # `roofline`); all stages are timed in the serial pass below
        _C.set_stage_timing(2 if stage_timing else 0)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        # Frames are independent, so the loop alternates over `--streams` HIP streams: the
        # VALU/latency-bound render of frame k overlaps the HBM-bound binning of frame k+1 and the
        # one host round trip per frame (num_rendered) no longer idles the GPU.
        streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]
        nthreads = max(1, min(args.host_threads, len(streams)))
        t0 = time.perf_counter()
        works = []
        GB = max(1, args.gather_batch)
        if nthreads == 1:
            for s in range(K):
                with torch.cuda.stream(streams[s % len(streams)]):
                    tj.pack_u8(render_frame(frames_of(s)), out=local[s])
                # RCCL only for the image gather, issued per batch of frames so that it travels over
                # xGMI while the next batch renders; only the last batch's transfer is exposed
                if world > 1 and args.gather_batch > 0 and ((s + 1) % GB == 0 or s == K - 1):
                    b0 = (s // GB) * GB
                    for st_ in streams:
                        torch.cuda.current_stream().wait_stream(st_)
                    src = local[b0:s + 1] if backend == "nccl" else local[b0:s + 1].cpu()
                    works.append(dist.gather(
                        src, gather_list=[g[b0:s + 1] for g in gather_bufs] if rank == 0 else None,
                        dst=0, async_op=True))
        else:
            import threading

            def issue(t):
                torch.cuda.set_device(dev)
                mine = streams[t::nthreads]
                for n, s in enumerate(range(t, K, nthreads)):
                    with torch.cuda.stream(mine[n % len(mine)]):
                        tj.pack_u8(render_frame(frames_of(s)), out=local[s])

            workers = [threading.Thread(target=issue, args=(t,)) for t in range(nthreads)]
            for w in workers:
                w.start()
            for w in workers:
                w.join()
        for st_ in streams:
            torch.cuda.current_stream().wait_stream(st_)
        if world > 1:
            if works:
                for w_ in works:
                    w_.wait()
            else:   # one gather at the end (--gather-batch 0, or the multi-threaded issue loop)
                dist.gather(local if backend == "nccl" else local.cpu(), gather_list=gather_bufs, dst=0)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        stage_sum, ncalls = _C.stage_timing() if stage_timing else ([0.0] * 8, 0)
        # short serial (one stream) pass: per-stage durations without the interference of the
        # overlapped frames, reported beside the timed-region figures
        iso_sum, iso_calls = [0.0] * 8, 0
        if stage_timing:
            _C.set_stage_timing(1)
            for s in range(min(K, 20)):
                tj.pack_u8(render_frame(frames_of(s)))
            torch.cuda.synchronize()
            iso_sum, iso_calls = _C.stage_timing()
        _C.set_stage_timing(0)

        # Host-delivered frames: the same loop, but every frame is packed to
        # rgb8 on the device and lands in pinned host memory (3 B/pixel over PCIe), consumed one
        # frame behind the producer.  Reported beside the headline figure, never as `value`.
        delivery = None
        if world == 1 and not args.no_delivery:
            nd = K
            fd = tj.FrameDelivery(H, W, depth=2 * len(streams), truncate=True)
            torch.cuda.synchronize()
            td0 = time.perf_counter()
            checksum = 0
            tickets = []
            for s in range(nd):
                with torch.cuda.stream(streams[s % len(streams)]):
                    tickets.append(fd.submit(render_frame(frames_of(s))))   # pack clamps
                if s >= len(streams):
                    checksum += int(fd.get(tickets[s - len(streams)])[0, 0, 0])
            for s in range(max(0, nd - len(streams)), nd):
                checksum += int(fd.get(tickets[s])[0, 0, 0])
            td1 = time.perf_counter()
            delivery = {"frames_per_s": nd / (td1 - td0), "frames": nd,
                        "bytes_per_frame": 3 * H * W,
                        "what": "rgb8 [H,W,3] in pinned host memory (device pack + async D2H), "
                                "consumer one frame behind"}

        elapsed = t1 - t0
        if world > 1:
            te = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            elapsed = float(te.item())

        # untimed statistics pass: V and R of the frames this rank rendered
        Vs, Rs = [], []
        e = torch.Tensor([])
        sem0 = torch.zeros(P, 0, device=dev)
        for s in range(min(K, 50)):
            rs = rasterizers[frames_of(s) % NUM_FRAMES].raster_settings
            out = _C.rasterize_gaussians(rs.bg, sc.means3D, e, sem0, sc.opacity, sc.scales,
                                         sc.rotations, rs.scale_modifier, e, rs.viewmatrix,
                                         rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height,
                                         rs.image_width, sc.shs, rs.sh_degree, rs.campos, False, False)
            Rs.append(int(out[0]))
            Vs.append(int((out[5] > 0).sum()))
        torch.cuda.synchronize()

    if rank == 0:
        T_tiles = ((W + 15) // 16) * ((H + 15) // 16)
        N = W * H
        R_avg = sum(Rs) / len(Rs)
        V_avg = sum(Vs) / len(Vs)
        fps = world * K / elapsed
        ms_per_step = 1000.0 * elapsed / K
        names = ["preprocess", "depth_sort", "offsets_scan_and_readback", "emit", "tile_sort",
                 "tile_ranges", "render", "semantic_render"]
        # the timed region records the render stage only (see set_stage_timing(2) above)
        stages = {n: (stage_sum[i] / ncalls if ncalls and (stage_sum[i] > 0 or n == "render") else None)
                  for i, n in enumerate(names)}
        stages_iso = {n: (iso_sum[i] / iso_calls if iso_calls else None) for i, n in enumerate(names)}
        render_ms = stages["render"]
        b_render = 44.0 * R_avg + 8.0 * T_tiles + 20.0 * N
        b_frame = P * (48 + 12 * M) + 40.0 * V_avg + 88.0 * R_avg + 16.0 * T_tiles + 20.0 * N
        # HBM bytes per launch of the dominant kernel from the PMC passes of tools/profile_gpu.sh
        # (FETCH_SIZE / WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes), if a profile of
        # this same workload has been committed; otherwise null.
        traffic = None
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "round1_traffic.json")))
            wl = tr.get("workload", {})
            if wl.get("P") == P and wl.get("width") == W and wl.get("height") == H:
                traffic = tr["render_forward_kernel"]["hbm_bytes_corrected"]
        except Exception:
            traffic = None
        roof = None
        if render_ms:
            ach = b_render / (render_ms * 1e-3) / 1e9
            roof = {"kernel": "render_forward_kernel", "bound": "hbm", "achieved": ach,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": traffic, "algorithmic_bytes_per_launch": b_render,
                    "avg_launch_ms": render_ms,
                    "note": "render is VALU-bound (about 25 flop per pixel-splat pair), see DESIGN.md §6"}
        ach_f = b_frame / (ms_per_step * 1e-3) / 1e9
        line = {
            "metric": "frames/sec @1920x1280, ~2M Gaussians; achieved HBM GB/s vs peak",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[2]: scene-002-like synthetic street scene (seed %d), "
                                   "forward op + clamp + uint8 pack per frame, %d-pose drive, "
                                   "frames sharded round-robin over ranks" % (SCENE_SEED, NUM_FRAMES),
                       "P": P, "V_avg": V_avg, "R_avg": R_avg, "T": T_tiles, "width": W,
                       "height": H, "sh_degree": sc.sh_degree, "M": M, "S": 0,
                       "streams_per_gpu": max(1, args.streams),
                       "gather_batch_frames": args.gather_batch if world > 1 else None,
                       "host_threads_per_gpu": max(1, min(args.host_threads, max(1, args.streams))),
                       "parallelism": "replicas x%d, frame-sharded, final uint8 gather" % world},
            "roofline": roof,
            "frame_roofline": {"bound": "hbm", "achieved": ach_f, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": ach_f / HBM_PEAK_GBS,
                               "algorithmic_bytes_per_frame": b_frame},
            "stages_ms": stages,
            "stages_ms_serial": stages_iso,
            "delivery": delivery,
        }
        if stages_iso["render"]:
            ach_i = b_render / (stages_iso["render"] * 1e-3) / 1e9
            line["roofline_serial"] = {"kernel": "render_forward_kernel", "bound": "hbm",
                                       "achieved": ach_i, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": ach_i / HBM_PEAK_GBS,
                                       "avg_launch_ms": stages_iso["render"],
                                       "note": "same kernel timed with one stream (no frame overlap)"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(int(R_avg))
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
