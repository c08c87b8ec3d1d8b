"""Host-side invariants of the blend-checkpoint layout (csrc/common.h CK_*), checked
on the CPU with the constants read out of the header: the slot ranges of the long tiles of a frame
never overlap, fit the carved region, and the backward's item bound covers every item the forward
can publish."""
import os
import re

import numpy as np

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "gaussianrpg_amd", "csrc", "common.h")


def _constants():
    src = open(HDR).read()
    seg = int(re.search(r"#define GRPG_CK_SEG (\d+)", src).group(1))
    m = re.search(r"#define GRPG_CK_LONG_MIN \((\d+) \* GRPG_CK_SEG\)", src)
    long_min = int(m.group(1)) * seg
    # the formulas this test restates must still be the header's
    assert "(3ull * range_x) / (2ull * CK_SEG)" in src and "len / CK_SEG + 1u" in src
    assert "(3ull * Rcap) / (2ull * CK_SEG)) + 2u" in src
    return seg, long_min


def _base(x, seg):
    return (3 * x) // (2 * seg)


def test_slot_ranges_do_not_overlap_and_fit():
    seg, long_min = _constants()
    assert long_min >= 2 * seg
    rng = np.random.default_rng(0)
    for trial in range(200):
        T = int(rng.integers(1, 400))
        # a mix of empty, short, just-long and very long lists, in tile order (ranges are a prefix sum)
        kind = rng.integers(0, 5, T)
        lens = np.where(kind == 0, 0, np.where(kind == 1, rng.integers(1, long_min, T),
                np.where(kind == 2, long_min + rng.integers(0, 3, T),
                         np.where(kind == 3, rng.integers(long_min, 8 * long_min, T), rng.integers(0, 40, T)))))
        starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
        R = int(lens.sum())
        Rcap = R + int(rng.integers(0, 1000))
        slots = (3 * Rcap) // (2 * seg) + 2
        prev_end, items = 0, 0
        for x, n in zip(starts.tolist(), lens.tolist()):
            if n < long_min:
                continue
            cap = n // seg + 1
            b = _base(x, seg)
            assert b >= prev_end, "slot ranges of consecutive long tiles overlap"
            assert b + cap <= slots, "records beyond the carved region"
            prev_end = b + cap
            items += cap
        # the backward launches ckpt_slots(R) item workgroups for num_rendered = R
        assert items <= (3 * R) // (2 * seg) + 2


def test_checkpoint_cut_points_bound_the_record_count():
    """A quarter stores a record when a batch ends >= CK_SEG positions behind the previous cut (first
    cut at >= CK_SEG), plus the final one: never more than len / CK_SEG + 1 records."""
    seg, long_min = _constants()
    rng = np.random.default_rng(1)
    for _ in range(300):
        n = int(rng.integers(long_min, 20 * long_min))
        pos, nxt, k = 0, seg, 0
        while pos < n:
            pos = min(n, pos + int(rng.integers(1, 400)))   # batch ends anywhere
            if pos >= nxt:
                k += 1
                nxt = pos + seg
        assert k + 1 <= n // seg + 1
