"""The algebra behind render_bwd.hip's per-pixel state (round 6), checked in float64 on the CPU.

The reference's blend backward (cuda_rasterizer/backward.cu:556-614) carries, per pixel and per output
channel (3 colours, depth, accumulated alpha, S semantics), an accumulator of what lies BEHIND the current
splat, and uses the accumulators in one sum only:

    dL_dopa = T * sum_ch (c_ch - accum_rec_ch) * dL_ch  +  (-T_final / (1 - alpha)) * (bg . dL_dpixel)

The HIP kernel carries ONE scalar A = sum_ch accum_rec_ch * dL_ch (colour, depth, semantics), and no state at
all for the accumulated alpha, whose term T (1 - accum_alpha_rec) dL_dalpha equals (T_final / (1 - alpha))
dL_dalpha and joins the background term.  Both identities are exact in real arithmetic; this test walks random
splat sequences back to front with the reference's per-channel recurrences (as oracle/gs_oracle.c restates
them) and with the scalar form, and compares dL_dopa per splat."""
import numpy as np
import pytest


def _walk_reference(alpha, col, dL, dLa, bg_dot, T_final):
    """backward.cu:545-614 for one pixel; col [n, C], dL [C]; returns dL_dopa per splat (back to front order)."""
    n, C = col.shape
    T = T_final
    accum = np.zeros(C)
    accum_a = 0.0
    last_alpha = 0.0
    last_col = np.zeros(C)
    out = np.zeros(n)
    for i in range(n - 1, -1, -1):
        a = alpha[i]
        T = T / (1.0 - a)
        accum = last_alpha * last_col + (1.0 - last_alpha) * accum            # :561
        last_col = col[i].copy()
        d = float(((col[i] - accum) * dL).sum())                              # :565,:584,:594
        accum_a = last_alpha + (1.0 - last_alpha) * accum_a                   # :599
        d += (1.0 - accum_a) * dLa                                            # :600
        d *= T                                                                # :602
        last_alpha = a
        d += (-T_final / (1.0 - a)) * bg_dot                                  # :611-614
        out[i] = d
    return out


def _walk_scalar(alpha, col, dL, dLa, bg_dot, T_final):
    """render_bwd.hip backward_rect: A = sum_ch acc_ch dL_ch, the alpha plane inside the background term."""
    n, C = col.shape
    T = T_final
    A = 0.0
    nTfBg = T_final * (dLa - bg_dot)
    out = np.zeros(n)
    for i in range(n - 1, -1, -1):
        a = alpha[i]
        inv = 1.0 / (1.0 - a)
        T = T * inv
        cD = float((col[i] * dL).sum())
        d = cD - A
        A = A + a * d
        out[i] = inv * nTfBg + d * T
    return out


@pytest.mark.parametrize("seed", range(6))
def test_scalar_accumulator_equals_per_channel_recurrences(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 400))
    C = int(rng.choice([4, 6, 19]))                       # colour + depth (+ semantics)
    alpha = np.clip(rng.uniform(1.0 / 255.0, 1.2, n), None, 0.99)
    col = rng.uniform(0.0, 1.0, (n, C))
    col[:, 3] = rng.uniform(0.5, 300.0, n)                # depth channel
    dL = rng.normal(size=C)
    dLa = float(rng.normal())
    bg_dot = float(rng.normal())
    T_final = float(np.prod(1.0 - alpha))
    ref = _walk_reference(alpha, col, dL, dLa, bg_dot, T_final)
    new = _walk_scalar(alpha, col, dL, dLa, bg_dot, T_final)
    scale = np.abs(ref).max() + 1e-300
    assert np.abs(ref - new).max() / scale < 1e-9


def test_accumulated_alpha_term_is_the_background_factor():
    """T_i (1 - accum_alpha_rec_i) == T_final / (1 - alpha_i): what lies behind splat i lets through
    prod_{j > i} (1 - alpha_j), and T_i times that times (1 - alpha_i) is the final transmittance.
    (Short, translucent lists: 1 - accum_alpha_rec cancels once the splats behind are opaque -- in the
    recurrence's form the term is then lost to rounding, in the product's form it is not.)"""
    rng = np.random.default_rng(7)
    for _ in range(50):
        alpha = rng.uniform(0.004, 0.5, int(rng.integers(1, 16)))
        T_final = np.prod(1.0 - alpha)
        T, acc_a, last = T_final, 0.0, 0.0
        for i in range(len(alpha) - 1, -1, -1):
            T = T / (1.0 - alpha[i])
            acc_a = last + (1.0 - last) * acc_a
            last = alpha[i]
            want = T_final / (1.0 - alpha[i])
            assert abs(T * (1.0 - acc_a) - want) <= 1e-10 * want
