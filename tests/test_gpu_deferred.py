"""GPU tests (-m gpu) of the deferred-count frame path (grpg_forward_deferred / grpg_frame_status,
GaussianRasterizer.forward_deferred, trajectory.DeferredFrames): frames enqueued without the
per-frame wait for num_rendered are bit-identical to the synchronous entry point's, a frame that
outgrows the remembered capacity is reported (never silently wrong) and rendered again, and the
status call behaves on tickets it does not know."""
import numpy as np
import pytest
import torch

from gaussianrpg_amd import harness as hz
from gaussianrpg_amd import trajectory as tj

pytestmark = pytest.mark.gpu


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


def test_deferred_frames_equal_synchronous_frames(dev):
    from gaussianrpg_amd.rasterizer import _C, frame_ok
    _C.reset_capacity_hints()
    W, H = 640, 384
    d = hz.street_scene(60000, seed=8).to(dev)
    with torch.no_grad():
        for k in range(6):      # frame 0 has no capacity history: it runs synchronously inside
            r = _rasterizer(dev, d, k, W, H)
            ticket, color, radii, depth, alpha, sem = r.forward_deferred(**_inputs(d))
            assert frame_ok(ticket) is True
            ref = r(means2D=None, **_inputs(d))
            torch.cuda.synchronize()
            for got, want in zip((color, radii, depth, alpha), ref[:4]):
                assert torch.equal(got, want)


def test_many_frames_in_flight_and_late_status(dev):
    """Eight frames are enqueued before any status is asked for (two streams), then checked out of
    order; a second query of a resolved ticket gives the same answer."""
    from gaussianrpg_amd.rasterizer import _C, frame_ok
    _C.reset_capacity_hints()
    W, H = 480, 320
    d = hz.street_scene(40000, seed=9).to(dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    with torch.no_grad():
        ref = [_rasterizer(dev, d, k, W, H)(means2D=None, **_inputs(d))[0].clone() for k in range(8)]
        torch.cuda.synchronize()
        out = []
        for k in range(8):
            with torch.cuda.stream(streams[k % 2]):
                out.append(_rasterizer(dev, d, k, W, H).forward_deferred(**_inputs(d)))
        for k in (5, 0, 7, 2, 1, 6, 3, 4):
            assert frame_ok(out[k][0]) is True
            assert frame_ok(out[k][0]) is True
        torch.cuda.synchronize()
        for k in range(8):
            assert torch.equal(out[k][1], ref[k])


def test_overflowing_frame_is_reported_and_rendered_again(dev):
    """Same (P, W, H) key, far more instances: the deferred frame is enqueued for the small
    scene's capacity, its status says so, DeferredFrames renders it again."""
    from gaussianrpg_amd.rasterizer import _C, frame_ok
    import os
    if os.environ.get("GRPG_SYNC_R", "0") not in ("", "0"):
        pytest.skip("GRPG_SYNC_R=1: exact binning mode, a deferred frame runs synchronously and cannot overflow")
    W, H, P = 640, 384, 6000
    small = hz.toy_scene(P, seed=5, sh_degree=1, depth=30.0).to(dev)
    big = hz.toy_scene(P, seed=5, sh_degree=1, depth=1.5).to(dev)
    r = _rasterizer(dev, small, 0, W, H)
    with torch.no_grad():
        want = r(means2D=None, **_inputs(big))[0].clone()
        torch.cuda.synchronize()
        for alg in (0, 1):
            _C.set_binning_algorithm(alg)
            try:
                _C.reset_capacity_hints()
                r(means2D=None, **_inputs(small))            # remembers the small capacity
                ticket = r.forward_deferred(**_inputs(big))[0]
                assert frame_ok(ticket) is False             # reported, not silently wrong
                r(means2D=None, **_inputs(big))              # the repeat: synchronous, records the capacity
                ticket, color = r.forward_deferred(**_inputs(big))[:2]   # which the next frame fits
                assert frame_ok(ticket) is True
                torch.cuda.synchronize()
                assert torch.equal(color, want)
                # the helper does the same on its own
                _C.reset_capacity_hints()
                r(means2D=None, **_inputs(small))
                got = torch.empty(3, H, W, dtype=torch.uint8, device=dev)
                frames = tj.DeferredFrames(window=4)
                frames.render(r, lambda c: tj.pack_u8(c, out=got), **_inputs(big), cov3D_precomp=None,
                              semantics=None)
                assert frames.finish() == 1
                torch.cuda.synchronize()
                assert torch.equal(got, tj.pack_u8(want))
            finally:
                _C.set_binning_algorithm(1)
                _C.reset_capacity_hints()


def test_frame_status_rejects_unknown_and_expired_tickets(dev):
    from gaussianrpg_amd.rasterizer import _C, frame_ok
    with pytest.raises(RuntimeError):
        _C.frame_status(10 ** 6, True)
    with pytest.raises(RuntimeError):
        _C.frame_status(-1, True)
    d = hz.toy_scene(500, seed=3, sh_degree=1).to(dev)
    r = _rasterizer(dev, d, 0, 64, 48)
    with torch.no_grad():
        first = r.forward_deferred(**_inputs(d))[0]
        assert frame_ok(first) is True
        for _ in range(64):                      # the ring of 64 slots comes round
            last = r.forward_deferred(**_inputs(d))[0]
        assert frame_ok(last) is True
        with pytest.raises(RuntimeError):
            _C.frame_status(first, True)         # its slot belongs to a later frame now
