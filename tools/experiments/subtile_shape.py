"""Experiment (CPU, numpy): would 8x8 quadrants instead of 16x4 strips as the sub-tile of a wave
lower the number of (wave, splat) evaluations of render / backward?  Brute force over the pixel
centres of sampled (tile, instance) pairs of the bench frame, float64, the reference's accept test
(forward.cu:410-420).  Saturation / early termination is ignored (it shortens both alike).
Reuses the sampling of tests/mask_check.py; its brute-force function is wrapped here, in this
process only, to tally the other shapes from the same accept matrices.
usage: python tools/experiments/subtile_shape.py [P] [samples]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import mask_check as mc   # noqa: E402

tally = {"n": 0, "strips": 0, "squares": 0, "eighths": 0, "accepted": 0, "tiles_hit": 0}
_orig = mc.truth_mask


def _accept(gx, gy, a, b, c, op, X0, Y0):
    xs = np.arange(16)[None, :, None]; ys = np.arange(16)[None, None, :]
    dx = gx[:, None, None] - (X0[:, None, None] + xs)
    dy = gy[:, None, None] - (Y0[:, None, None] + ys)
    power = -0.5 * (a[:, None, None] * dx * dx + c[:, None, None] * dy * dy) - b[:, None, None] * dx * dy
    alpha = np.minimum(0.99, op[:, None, None] * np.exp(power))
    return (power <= 0) & (alpha >= 1.0 / 255.0)   # [n, x, y]


def _tallying_truth_mask(*args):
    acc = _accept(*args)
    n = acc.shape[0]
    strips = sum(acc[:, :, 4 * q:4 * q + 4].any(axis=(1, 2)).sum() for q in range(4))
    squares = sum(acc[:, 8 * i:8 * i + 8, 8 * j:8 * j + 8].any(axis=(1, 2)).sum() for i in range(2) for j in range(2))
    eighths = sum(acc[:, 8 * i:8 * i + 8, 4 * q:4 * q + 4].any(axis=(1, 2)).sum() for i in range(2) for q in range(4))
    tally["n"] += n; tally["strips"] += int(strips); tally["squares"] += int(squares)
    tally["eighths"] += int(eighths); tally["accepted"] += int(acc.sum())
    tally["tiles_hit"] += int(acc.any(axis=(1, 2)).sum())
    return _orig(*args)


if __name__ == "__main__":
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    nsamp = int(sys.argv[2]) if len(sys.argv) > 2 else 300_000
    mc.truth_mask = _tallying_truth_mask   # this process only
    try:
        mc.run(P, nsamp, verbose=False)
    finally:
        mc.truth_mask = _orig
    t = tally
    print("sampled (tile, instance) pairs: %d, of which reach a pixel at all: %d" % (t["n"], t["tiles_hit"]))
    for name, k, px in (("16x4 strips (now)", "strips", 64), ("8x8 squares", "squares", 64), ("8x4 eighths", "eighths", 32)):
        print("%-20s units hit per pair %.3f   lane utilisation %.3f   lane-evaluations per pair %.1f" % (
            name, t[k] / t["n"], t["accepted"] / (px * max(t[k], 1)), px * t[k] / t["n"]))
    print("squares / strips = %.3f" % (t["squares"] / max(t["strips"], 1)))
