"""Test infrastructure (CPU): validate the sub-tile reach masks that emit computes
(csrc/blend_math.h: quarter_reach_mask) against brute force over the pixel centres, on the bench
frame.  Run directly for the full-size check, or through tests/test_mask_conservative.py.

The mask must be CONSERVATIVE: a quarter that holds a pixel which passes the reference's alpha
test (forward.cu:410-420) must never be dropped.  Emulates the kernel's fp32 arithmetic with numpy
float32 and compares with float64 truth on a sample of tile instances; also reports how tight the
masks are.   usage: python tests/mask_check.py [P] [samples]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianrpg_amd import harness as hz          # noqa: E402
from oracle import torch_splat as ts                # noqa: E402

f32 = np.float32


def reach_mask_f32(gx, gy, a, b, c, op, X0, Y0):
    """numpy-float32 emulation of quarter_reach_mask(); returns uint8 masks."""
    gx, gy, a, b, c, op = (v.astype(f32) for v in (gx, gy, a, b, c, op))
    X0 = X0.astype(f32); Y0 = Y0.astype(f32)
    t = f32(2.0) * np.log(f32(255.0) * op)
    t = t + f32(1e-4) * np.abs(t) + f32(2e-3)
    p = b * b
    e = (b.astype(np.float64) * b.astype(np.float64) - p.astype(np.float64)).astype(f32)   # fma(b,b,-p)
    det = ((a.astype(np.float64) * c.astype(np.float64) - p.astype(np.float64)).astype(f32)) - e
    sane = (a > 0) & (c > 0) & (det > 0) & np.isfinite(t) & np.isfinite(det)
    xl = X0 - gx; xh = xl + f32(15.0)
    ytop = np.sqrt(np.maximum(t * a / det, f32(0)))
    xtop = -(b / a) * ytop
    xet = np.clip(xtop, xl, xh); xeb = np.clip(-xtop, xl, xh)
    ct = c * t
    dt_ = ct - det * xet * xet
    db_ = ct - det * xeb * xeb
    ic = f32(1.0) / c
    yup = (-b * xet + np.sqrt(np.maximum(dt_, f32(0)))) * ic
    ylo = (-b * xeb - np.sqrt(np.maximum(db_, f32(0)))) * ic
    yup = yup + (f32(1e-3) + f32(1e-5) * np.abs(yup))
    ylo = ylo - (f32(1e-3) + f32(1e-5) * np.abs(ylo))
    empty = (dt_ < 0) & (db_ < 0)
    transparent = op < f32((1.0 / 255.0) * 0.999)
    m = np.zeros(gx.shape, np.uint8)
    for q in range(4):
        yl_q = Y0 + f32(4 * q) - gy
        yh_q = yl_q + f32(3.0)
        hit = (ylo <= yh_q) & (yup >= yl_q) & ~empty
        hit = np.where(sane, hit, True) & ~transparent
        m |= (hit.astype(np.uint8) << q)
    return m


def truth_mask(gx, gy, a, b, c, op, X0, Y0):
    """float64 brute force over the 16x16 pixel centres: which quarters hold an accepted pixel."""
    n = gx.shape[0]
    xs = np.arange(16)[None, :, None]; ys = np.arange(16)[None, None, :]
    dx = gx[:, None, None] - (X0[:, None, None] + xs)
    dy = gy[:, None, None] - (Y0[:, None, None] + ys)
    power = -0.5 * (a[:, None, None] * dx * dx + c[:, None, None] * dy * dy) - b[:, None, None] * dx * dy
    alpha = np.minimum(0.99, op[:, None, None] * np.exp(power))
    acc = (power <= 0) & (alpha >= 1.0 / 255.0)          # [n, x, y]
    m = np.zeros(n, np.uint8)
    for q in range(4):
        m |= (acc[:, :, 4 * q:4 * q + 4].any(axis=(1, 2)).astype(np.uint8) << q)
    return m


def run(P=2_000_000, nsamp=1_500_000, verbose=True):
    """Returns (false-miss quarter bits, truth survive fraction, mask survive fraction)."""
    sc = hz.street_scene(P, seed=2)
    cam = hz.trajectory_camera(0)
    kw = hz.settings_kwargs(cam, 1)
    pre = ts.preprocess(sc.means3D, sc.opacity, image_height=kw["image_height"], image_width=kw["image_width"],
                        tanfovx=kw["tanfovx"], tanfovy=kw["tanfovy"], viewmatrix=kw["viewmatrix"],
                        projmatrix=kw["projmatrix"], campos=kw["campos"], sh_degree=1, shs=sc.shs,
                        scales=sc.scales, rotations=sc.rotations)
    minx, miny, maxx, maxy = (v.numpy() for v in pre["rect"])
    tiles = pre["tiles_touched"].numpy()
    vis = np.nonzero(tiles > 0)[0]
    counts = tiles[vis]
    R = int(counts.sum())
    rng = np.random.default_rng(0)
    pick = np.sort(rng.choice(R, size=min(nsamp, R), replace=False))
    starts = np.cumsum(counts) - counts
    owner = np.searchsorted(starts, pick, side="right") - 1
    gid = vis[owner]
    local = pick - starts[owner]
    w = (maxx - minx)[gid]
    tx = minx[gid] + local % w
    ty = miny[gid] + local // w
    m2 = pre["means2D"].numpy().astype(np.float64)
    con = pre["conic"].numpy().astype(np.float64)
    op = pre["opacity"].numpy().astype(np.float64)
    args = (m2[gid, 0], m2[gid, 1], con[gid, 0], con[gid, 1], con[gid, 2], op[gid],
            (tx * 16).astype(np.float64), (ty * 16).astype(np.float64))
    got = np.zeros(len(gid), np.uint8); ref = np.zeros(len(gid), np.uint8)
    for s in range(0, len(gid), 200_000):
        sl = slice(s, s + 200_000)
        a = tuple(v[sl] for v in args)
        got[sl] = reach_mask_f32(*a)
        ref[sl] = truth_mask(*a)
    missed = ref & ~got
    pc = lambda m: np.unpackbits(m[:, None], axis=1)[:, 4:].sum()   # noqa: E731
    if verbose:
        print("R=%d sampled=%d" % (R, len(gid)))
        print("FALSE MISSES (must be 0): %d quarter bits in %d instances" % (pc(missed), int((missed != 0).sum())))
        print("quarter survive: truth %.4f  mask %.4f   tile survive: truth %.4f mask %.4f" % (
            pc(ref) / (4.0 * len(gid)), pc(got) / (4.0 * len(gid)), (ref != 0).mean(), (got != 0).mean()))
    return int(pc(missed)), pc(ref) / (4.0 * len(gid)), pc(got) / (4.0 * len(gid))


if __name__ == "__main__":
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    nsamp = int(sys.argv[2]) if len(sys.argv) > 2 else 1_500_000
    sys.exit(0 if run(P, nsamp)[0] == 0 else 1)
