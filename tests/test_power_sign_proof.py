"""CPU test (-m "not gpu"): the claim behind the render's skipped `power > 0` compare (round 5).

`blend_math.h:splat_power_never_positive` lets a batch of splats run the quad loop without the reference's
`power > 0` test (cuda_rasterizer/forward.cu:420).  That is only legitimate if the device's fp32 evaluation of
power2 can NEVER come out positive for a conic that passes -- for any pixel offset.  The header argues it on paper
(rounding terms 60x below the margin); here the device's exact operation sequence is replayed in numpy fp32 on
200 k conics x 40 adversarial offset sets (along the major axis, where the quadratic cancels, at every scale from
1e-4 to 4000 px; integer pixel grids against float centres), and on conics far beyond the guard the same replay
DOES produce positive values -- the guard is what makes the short cut exact, not luck."""
import numpy as np

from helpers import device_power2, power_never_positive


def _conics(rng, n, lo, hi, smin_hi):
    ratio = 10 ** rng.uniform(lo, hi, n)
    smin = rng.uniform(0.55, smin_hi, n)   # the +0.3 dilation of the 2-D covariance keeps every axis >= 0.55 px
    smaj = smin * ratio
    th = rng.uniform(0, np.pi, n)
    c, s = np.cos(th), np.sin(th)
    l1, l2 = 1 / smaj ** 2, 1 / smin ** 2
    conic = np.stack([c * c * l1 + s * s * l2, c * s * (l1 - l2), s * s * l1 + c * c * l2], 1).astype(np.float32)
    return conic, c, s, smin, ratio


def test_power2_is_never_positive_where_the_guard_says_so():
    rng = np.random.default_rng(5)
    n = 200_000
    conic, c, s, smin, ratio = _conics(rng, n, 0.0, 3.0, 5.0)
    safe = power_never_positive(conic)
    assert 0.7 < safe.mean() < 0.9 and ratio[safe].max() < 500 and ratio[~safe].min() > 200   # the guard cuts near 300 : 1
    worst = -np.inf
    for rep in range(40):
        t = rng.uniform(-4000, 4000, n) * (10 ** rng.uniform(-4, 0, n))
        u = rng.normal(0, 1, n) * smin * rng.uniform(0, 2, n)
        dx, dy = t * c - u * s, t * s + u * c
        if rep % 2:   # pixel centres are integers, splat centres are not
            off = rng.uniform(0, 2000, n)
            dx = np.round(dx + off) - off
            dy = np.round(dy)
        p = device_power2(conic[safe], dx.astype(np.float32)[safe], dy.astype(np.float32)[safe])
        worst = max(worst, float(p.max()))
    assert worst <= 0.0, worst
    # exact zero offsets: +-0, not > 0
    z = np.zeros(int(safe.sum()), np.float32)
    assert not (device_power2(conic[safe], z, z) > 0).any()


def test_the_guard_is_needed():
    """Needles far beyond the guard: the same replay finds positive power2 (which the reference's test rejects and
    the exact loop keeps rejecting) -- and none of them passes the guard."""
    rng = np.random.default_rng(7)
    n = 100_000
    conic, c, s, _, _ = _conics(rng, n, 3.0, 6.0, 2.0)
    assert not power_never_positive(conic).any()
    t = rng.uniform(-4000, 4000, n)
    u = rng.normal(0, 0.01, n)
    p = device_power2(conic, (t * c - u * s).astype(np.float32), (t * s + u * c).astype(np.float32))
    assert (p > 0).sum() > 1000


def test_degenerate_conics_fail_the_guard():
    bad = np.array([[0, 0, 0], [np.nan, 0, 1], [1, 0, np.inf], [-1, 0, 1], [1, 0, -1], [1, 2, 1], [1e-20, 0, 1e-20]],
                   np.float32)
    assert not power_never_positive(bad).any()
