"""CPU test (-m "not gpu"): the indexing argument of the fused coarse emit (csrc/binning.hip
emit_coarse_fused_kernel, round 5), replayed in numpy.

The kernel replaces the offsets scan's down-sweep: a workgroup that owns output slots [o0, o0 + 2048) scans the
sums of 2048-Gaussian blocks, takes the LAST block B0 whose prefix is <= o0, scans the counts of the 4096 Gaussians
from B0's first one on, and claims that the owners of all its slots lie among them (every Gaussian owns >= 1 slot).
Here: the same steps on random count arrays (ragged tails, counts of 1 everywhere, long runs of big counts, V not a
multiple of the block) against the naive expansion np.repeat(arange(V), counts); the GPU tests check the kernel's
output bit for bit, this one checks that the argument the kernel rests on has no corner the tests' scenes miss."""
import numpy as np
import pytest

BLOCK = 2048      # SC_CHUNK == EMIT_PER_BLOCK
WIN = 2 * BLOCK   # FUSED_WIN


def _model(counts):
    V = counts.size
    nblocks = max(1, (V + BLOCK - 1) // BLOCK) + 3            # (the table is sized for P, not V: empty blocks behind)
    padded = np.zeros(nblocks * BLOCK, np.int64)
    padded[:V] = counts
    sums = padded.reshape(nblocks, BLOCK).sum(1)
    spine = np.concatenate([[0], np.cumsum(sums)[:-1]])        # exclusive prefix of the block sums
    total = int(sums.sum())
    owners = []
    for o0 in range(0, total, BLOCK):
        o1 = min(total, o0 + BLOCK)
        B0 = int(np.nonzero(spine <= o0)[0][-1])               # the kernel's binary search
        G0 = B0 * BLOCK
        c = padded[G0:G0 + WIN] if G0 + WIN <= padded.size else np.concatenate([padded[G0:], np.zeros(G0 + WIN - padded.size, np.int64)])
        off = spine[B0] + np.concatenate([[0], np.cumsum(c)[:-1]])
        inside = c > 0
        lo = np.nonzero(inside & (off <= o0) & (o0 < off + c))[0]
        hi = np.nonzero(inside & (off <= o1 - 1) & (o1 - 1 < off + c))[0]
        assert lo.size == 1 and hi.size == 1, (o0, lo, hi)     # both window ends found inside the 4096
        lo, hi = int(lo[0]), int(hi[0])
        assert lo < BLOCK and hi - lo + 1 <= BLOCK + 1         # owner of o0 within B0; <= 2049 owners
        # the kernel's marking + max-scan: slot s is owned by the last window index j with off[j] <= s
        own = np.zeros(o1 - o0, np.int64)
        w = off[lo:hi + 1]
        j = np.arange(1, w.size)
        m = w[1:] < o1
        np.maximum.at(own, (w[1:][m] - o0).astype(np.int64), j[m])
        own = np.maximum.accumulate(own)
        owners.append(G0 + lo + own)
    return np.concatenate(owners) if owners else np.zeros(0, np.int64), total


@pytest.mark.parametrize("case", ["ones", "uniform", "big_runs", "ragged", "one_block", "single"])
def test_fused_emit_window_matches_the_naive_expansion(case):
    rng = np.random.default_rng(["ones", "uniform", "big_runs", "ragged", "one_block", "single"].index(case))
    if case == "ones":
        counts = np.ones(5 * BLOCK + 17, np.int64)
    elif case == "uniform":
        counts = rng.integers(1, 5, 7 * BLOCK + 901)
    elif case == "big_runs":       # screen-filling splats: hundreds of super-tiles each, between runs of ones
        counts = np.ones(6 * BLOCK + 3, np.int64)
        for s in (100, 2047, 2048, 4100, 9000):
            counts[s:s + 40] = rng.integers(300, 600, 40)
    elif case == "ragged":
        counts = rng.integers(1, 3, 3 * BLOCK - 1)
        counts[-1] = 5000           # the very last Gaussian owns several blocks of slots
    elif case == "one_block":
        counts = rng.integers(1, 4, 700)
    else:
        counts = np.array([3], np.int64)
    got, total = _model(counts)
    want = np.repeat(np.arange(counts.size), counts)
    assert total == want.size
    np.testing.assert_array_equal(got, want)
