"""Engine-wide switches.

align_corners: grid_sample semantics of every warp.  False = what the unmodified
reference executes under torch >= 1.3 (default, the as-run parity target); True = the authors'
torch-1.0 semantics.
"""
align_corners = False

# True: check every photometric term for NaN eagerly (AssertionError, one host sync per term, the
# reference's behaviour, loss_functions.py:60,105,115).  False: device flag, see loss_functions.check_finite().
strict_nan_checks = False

# True: run-to-run bit-reproducible training steps.  The one order-dependent reduction of the step -- the feature-gradient
# scatter of Back2Future's warps (float atomics, as in the reference's grid_sample backward) -- then accumulates 64-bit
# fixed-point integers instead (cc_feature_warp_bwd_det: four launches per warp instead of two, ~+0.3 ms per step).
deterministic = False
