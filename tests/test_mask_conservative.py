"""CPU: the quarter reach masks of emit (fp32 emulation of csrc/blend_math.h:quarter_reach_mask,
tests/mask_check.py) never drop a quarter that holds an accepted pixel, and stay tight."""
import mask_check


def test_quarter_masks_are_conservative_and_tight():
    missed, truth, mask = mask_check.run(P=300_000, nsamp=120_000, verbose=False)
    assert missed == 0                      # conservative: no false miss
    assert mask <= truth + 0.01             # tight: within one point of brute force
    assert 0.2 < truth < 0.6                # the 3-sigma square over-bins by 2-3x
