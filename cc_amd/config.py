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

# True: the four networks of a CCTrainer step run on HIP streams of their own (DispResNet6 and Back2Future on one side stream
# each, PoseNetB6 and MaskNet6 on the step's stream; forward and backward), joined before the losses / the optimizer.  Their layer
# chains are independent between the input frames and the losses (train.py:454-463 calls them one after the other), and most of
# their launches on the <= 32x104 levels leave CUs idle: as branches of the captured hipGraph they overlap (tools/overlap_probe.py:
# two chains of 208-416-workgroup convolutions, 1.69 -> 1.14 ms; the full step 20.6 -> 16.7 ms).  Same kernels, same arithmetic,
# same results.  True / 2: two side streams; 3: a third one for MaskNet6 (measured: no better); False / 0: one stream (rounds 1-4).
# Has no effect on CPU tensors (tests/hipemu).
net_streams = True


# True: in the per-network pipeline a network that marks points of its forward pass (DispResNet6: GRAD_CHUNKS) hands the parameters
# behind a mark to its gradient tail (all-reduce -> Adam -> weight images, on the step's origin stream, idle by then) as soon as its
# backward pass has come back to that mark, instead of the whole segment at the end.  DispResNet6 finishes its backward pass LAST, so
# its tail (0.47 ms on one GPU; data-parallel also its 219 MB exchange) is the one part of the optimizer nothing else covers.  OFF by
# default: on one GPU the two extra cross-stream dependencies cost the replayed graph more than the hidden tail saves (18.25 vs 16.69
# ms; on a fourth stream 19.15; profiles/r06_ab_round6.txt) -- a data-parallel run should measure both.
grad_chunks = False

# Pure bias-gradient sums (layers without an activation behind the bias) parked until the end of a backward stage and done by ONE
# table launch (cc_bias_grad_table), instead of one partial-sum pass per layer right where the gradient appears (whose second stage
# joins the stage's reduce table either way).  Built in round 3 for the single-stream step; with the networks on streams of their own
# the table launches sit in the streams' tails, where nothing overlaps them: same-box A/B 16.30 / 16.33 / 16.26 ms without against
# 16.42 / 16.40 / 16.35 / 16.38 with (profiles/r06_ab_round6.txt).  OFF.
bias_table = False

# The loss phase on two streams (cc_amd.trainer.cc_forward): consensus target + flow photometric loss on Back2Future's stream beside
# the rigid photometric loss and the mask / smoothness terms on the step's own.  HIP devices with config.net_streams only.
loss_stream = True


class _Debug:
    """A/B and diagnosis switches of the host glue.  The product reads nothing from the process environment: these are plain attributes that
    tools/ab_env.py (bench.py, the A/B scripts, tests/conftest.py) sets from CC_* variables explicitly.  All False / None = the
    shipped step."""
    no_slice_gy = False          # copy a concat gradient's channel slice instead of reading it in place
    no_wgrad_defer = False       # reduce every weight gradient right behind its kernel instead of once per backward stage
    no_sum_n = False             # pairwise adds for multi-consumer gradients instead of one n-ary sum
    no_wgrad_list = False        # the stage's last parked weight-gradient groups one by one
    no_wgrad_queue = False       # no parking of same-shaped weight gradients at all
    force_comm = False           # issue the gradient collectives on a one-rank process group too (tests, tools)
    capture_mode = "thread_local"    # hipGraph capture error mode of CCTrainer
    net_stream_priority = (0, 0, 0)  # HIP stream priorities of the networks' side streams (0 normal, -1 high)
    chunk_inline = False         # measurement: the gradient chunks' tails on the network's own stream instead of its tail stream
    pipe_extra = {}              # measurement: {network name: extra 256 MB fills (~50 us each) in its tail} -- how much slack has its stream?
    pipe_skip_tail = ()          # per-network pipeline, measurement only: names of the networks whose Adam segment + weight-image refresh are skipped
    reduce_trace = None          # a list: every weight-gradient reduce descriptor of the step is appended to it (tools/reduce_bytes.py)
    library_path = None          # another build of libccengine.so (the tools build): picked up by _lib.engine() on first use


debug = _Debug()
