"""CPU test (-m "not gpu") of trajectory.DeferredFrames' bookkeeping with a stand-in rasterizer:
statuses are read `window` frames behind the loop and the rest in finish(), a frame reported as
overflowed is rendered again through the synchronous entry point and consumed again, in place."""
import torch

from gaussianrpg_amd import rasterizer as rz
from gaussianrpg_amd import trajectory as tj


class _FakeRasterizer:
    """forward_deferred hands out tickets 0, 1, 2 ...; frames listed in `bad` come out wrong (zeros)
    and their status says so; the synchronous call always returns the right frame."""

    def __init__(self, bad, log):
        self.bad, self.log, self.next = set(bad), log, 0

    def _frame(self, value):
        return torch.full((3, 2, 2), float(value))

    def forward_deferred(self, value):
        t = self.next
        self.next += 1
        self.log.append(("deferred", t))
        color = torch.zeros(3, 2, 2) if t in self.bad else self._frame(value)
        return t, color, None, None, None, None

    def __call__(self, means2D=None, value=None):
        self.log.append(("sync", value))
        return (self._frame(value),)


def test_deferred_frames_window_and_redo(monkeypatch):
    log, asked = [], []
    rast = _FakeRasterizer(bad={2, 7}, log=log)

    def fake_frame_ok(ticket, wait=True):
        asked.append((ticket, len([e for e in log if e[0] == "deferred"])))
        return ticket not in rast.bad

    monkeypatch.setattr(rz, "frame_ok", fake_frame_ok)
    out = torch.zeros(10, 3, 2, 2)
    frames = tj.DeferredFrames(window=3)
    for k in range(10):
        frames.render(rast, lambda c, k=k: out[k].copy_(c), value=k + 1)
        assert len(frames.pending) <= 3
    # statuses are asked for 3 frames behind the loop: ticket t once t + 4 frames have been enqueued
    assert [t for t, _ in asked] == list(range(7))
    assert all(enq == t + 4 for t, enq in asked)
    assert frames.finish() == 2
    assert [t for t, _ in asked] == list(range(10)) and not frames.pending
    for k in range(10):
        assert torch.equal(out[k], torch.full((3, 2, 2), float(k + 1)))     # the bad ones were redone
    assert [e for e in log if e[0] == "sync"] == [("sync", 3), ("sync", 8)]
