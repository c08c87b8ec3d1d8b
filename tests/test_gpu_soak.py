"""-m gpu: a few thousand calls through every entry point, interleaved the way a long job mixes them
(evaluation frames of five shapes, deferred frames on three streams, training steps whose backward
comes late or never) -- the library's per-shape capacity hints, its table of geometry blobs a
backward may still claim, the deferred-frame tickets and the pinned count words are all bounded
tables that recycle; nothing may grow, nothing may go stale:

* device memory held by the caching allocator and the library's own allocations stay flat;
* the same input renders the same bits at the end as at the start;
* a backward that comes after 200 other forwards still finds its forward's blobs intact;
* forwards whose backward never comes cost nothing later.
"""
import pytest
import torch

from gaussianrpg_amd import harness as hz
from gaussianrpg_amd import trajectory as tj

pytestmark = pytest.mark.gpu

SHAPES = [(640, 384, 30000), (333, 217, 9000), (96, 64, 1500), (1024, 640, 60000), (480, 320, 20000)]


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X (no ROCm device visible)")
    return torch.device("cuda:0")


def _rasterizer(dev, sc, k, W, H):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    cam = hz.trajectory_camera(k, W=W, H=H, device=dev)
    return GaussianRasterizer(GaussianRasterizationSettings(
        **hz.settings_kwargs(cam, sc.sh_degree, bg=torch.zeros(3, device=dev))))


def _inputs(d):
    return dict(means3D=d.means3D, opacities=d.opacity, shs=d.shs, scales=d.scales, rotations=d.rotations)


def _free_device_bytes():
    free, _total = torch.cuda.mem_get_info()
    return free


def test_mixed_entry_points_for_thousands_of_calls(dev):
    from gaussianrpg_amd.rasterizer import _C
    _C.reset_capacity_hints()
    scenes = [hz.street_scene(P, seed=20 + i).to(dev) for i, (_, _, P) in enumerate(SHAPES)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]

    def eval_frame(i, k):
        W, H, _ = SHAPES[i]
        with torch.no_grad():
            return _rasterizer(dev, scenes[i], k, W, H)(means2D=None, **_inputs(scenes[i]))

    def train_forward(i, k):
        W, H, P = SHAPES[i]
        d = scenes[i]
        leaves = dict(means3D=d.means3D.clone().requires_grad_(True), opacities=d.opacity.clone().requires_grad_(True),
                      shs=d.shs.clone().requires_grad_(True), scales=d.scales.clone().requires_grad_(True),
                      rotations=d.rotations.clone().requires_grad_(True))
        m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
        out = _rasterizer(dev, d, k, W, H)(means2D=m2, **leaves)
        return leaves, out

    first = [tuple(t.clone() for t in eval_frame(i, 3)[:4]) for i in range(len(SHAPES))]
    # reference gradient of one training step, taken now, compared with a LATE backward below
    lv0, out0 = train_forward(0, 2)
    (out0[0].sum() + out0[2].sum()).backward()
    g_ref = lv0["means3D"].grad.clone()
    torch.cuda.synchronize()

    rounds, mem_marks = 12, []
    for rnd in range(rounds):
        # a training forward whose backward comes only after everything else of this round
        late_leaves, late_out = train_forward(0, 2)
        for it in range(100):
            i = it % len(SHAPES)
            eval_frame(i, it % 11)                         # synchronous evaluation frames
            if it % 4 == 0:                                # training steps, backward at once
                lv, out = train_forward(i, it % 7)
                (out[0].mean() + out[2].mean() + out[3].mean()).backward()
            if it % 10 == 0:                               # a forward whose backward never comes
                train_forward((i + 1) % len(SHAPES), it % 5)
        # deferred frames, three streams, one shape per stream, checked by the window
        with torch.no_grad():
            frames = tj.DeferredFrames(window=4)
            sink = [None]
            for it in range(150):
                i = it % 3
                W, H, _ = SHAPES[i]
                with torch.cuda.stream(streams[i]):
                    frames.render(_rasterizer(dev, scenes[i], it % 9, W, H),
                                  lambda color: sink.__setitem__(0, color), **_inputs(scenes[i]))
            frames.finish()
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        # the late backward: 100 evaluation forwards, 25 training steps, 10 abandoned forwards and 150
        # deferred frames after its forward
        (late_out[0].sum() + late_out[2].sum()).backward()
        torch.cuda.synchronize()
        assert torch.allclose(late_leaves["means3D"].grad, g_ref, rtol=1e-3, atol=1e-5 * float(g_ref.abs().max()))
        del late_leaves, late_out
        torch.cuda.synchronize()
        mem_marks.append((torch.cuda.memory_reserved(dev), _free_device_bytes()))

    # the same bits as at the start, for every shape
    for i in range(len(SHAPES)):
        again = eval_frame(i, 3)
        for a, b in zip(again[:4], first[i]):
            assert torch.equal(a, b)
    # flat after the first round (which sizes the allocator's pools and the library's tables):
    # neither torch's reserved pool nor the device's free memory (which also sees the library's own
    # hipMalloc / pinned allocations) moves by more than 64 MiB
    reserved = [m[0] for m in mem_marks[1:]]
    free = [m[1] for m in mem_marks[1:]]
    assert max(reserved) - min(reserved) <= 64 << 20, reserved
    assert max(free) - min(free) <= 64 << 20, free
