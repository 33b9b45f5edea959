"""CPU-side parity of the HIP kernel SOURCES: cc_amd/csrc/*.hip compiled for x86 against the fiber-based
HIP shim (tests/hipemu) and driven through cc_amd's real Python glue + C ABI, vs the oracle.  The same
cases run against the real gfx950 library in tests/test_kernels_gpu.py (-m gpu)."""
import pytest

import parity
from hipemu.emu import emulated_engine


@pytest.fixture(scope="module", autouse=True)
def _emu():
    with emulated_engine():
        yield


def test_warps():
    parity.check_warps("cpu")


def test_ssim():
    parity.check_ssim("cpu")


def test_losses():
    parity.check_losses("cpu")


def test_occluded_flow_loss():
    parity.check_occluded_photo_loss("cpu")


def test_pyramid():
    parity.check_pyramid("cpu")


def test_warps_bit_exact_vs_reference_golden(golden_dir):
    r = parity.check_warps_bit_exact_vs_golden("cpu", golden_dir)
    print(r)
    for tag, d in r.items():
        assert d["tap_flip_rate_device_P"] <= 2e-3, (tag, d)      # measured: see the printed dict
        for k in ("inverse_warp", "pose2flow", "flow_warp", "feature_warp", "grid"):
            assert d[k] == 1.0, (tag, k, d)


def test_losses_vs_reference_golden(golden_dir):
    print(parity.check_losses_vs_golden("cpu", golden_dir))
    print("measured against the bars:", parity.MEASURED)


def test_convs():
    parity.check_convs("cpu")


def test_convs_prepacked_weight_images():
    parity.check_convs("cpu", prepack=True)


def test_convs_winograd():
    parity.check_convs("cpu", cases=parity.CONV_CASES_WINO, tcases=[], prepack=True)      # the small maps run split-K


def test_convs_winograd_fused_epilogue(monkeypatch):
    monkeypatch.setenv("CC_WINO_MINQ", "1")              # every case on the Winograd kernel ...
    monkeypatch.setenv("CC_WINO_SPLIT_BELOW", "0")       # ... with the whole reduction in one workgroup: bias / residual / act in the kernel
    parity.check_convs("cpu", cases=parity.CONV_CASES_WINO, tcases=[], prepack=True)


@pytest.mark.parametrize("tile", [1, 2])
def test_convs_winograd_small_tile(monkeypatch, tile):
    # the 32 x 32 instances of the Winograd kernel (wino.hip k_wino_f2x3_s): four waves (two workgroups per CU) / eight waves with the
    # reduction halved inside the workgroup; every case with the whole reduction in the launch (fused epilogue), then sliced (split-K)
    monkeypatch.setenv("CC_WINO_MINQ", "1")
    monkeypatch.setenv("CC_WINO_SMALL", "2")
    monkeypatch.setenv("CC_WINO_S_TILE", str(tile))
    monkeypatch.setenv("CC_WINO_TRACE", "1")
    parity.check_convs("cpu", cases=parity.CONV_CASES_WINO, tcases=[], prepack=True)
    parity.check_conv_groups("cpu", cases=((2, 48, 16, 32, 64, 48, 1),))


@pytest.mark.parametrize("tile", [0, 1, 2])
def test_convs_winograd_nonsmooth_epilogues_unsplit(monkeypatch, tile):
    # ReLU / LeakyReLU (+ residual) in the fused forward epilogue and the deferred activation backward in the data-gradient epilogue
    # of launches that hold the whole reduction, with the activation derivative pinned (parity.check_convs_act_pinned): 64 x 64 blocks
    # and both 32 x 32 instances
    monkeypatch.setenv("CC_WINO_MINQ", "1")
    if tile == 0:
        monkeypatch.setenv("CC_WINO_SMALL", "0")
        monkeypatch.setenv("CC_WINO_SPLIT_BELOW", "0")
    else:
        monkeypatch.setenv("CC_WINO_SMALL", "2")
        monkeypatch.setenv("CC_WINO_S_TILE", str(tile))
        monkeypatch.setenv("CC_WINO_S_STAGE", "1")       # (cost model: never slice the reduction across workgroups)
        monkeypatch.setenv("CC_WINO_S_ALONE", "1")
    rep = parity.check_convs_act_pinned("cpu", cases=parity.CONV_CASES_WINO_ACT_SMALL)
    assert len(rep) == 3


def test_convs_winograd_weight_gradient(monkeypatch):
    # the small test maps on the Winograd F(3x3, 2x2) weight-gradient kernel (wino_wgrad.hip): odd heights, channel counts that are
    # not multiples of 64, tile rows that are not multiples of the 8-tile chunks, several splits
    monkeypatch.setenv("CC_WW_MINQ", "1")
    monkeypatch.setenv("CC_WW_MINM", "1")
    monkeypatch.setenv("CC_WW_MINC", "1")
    monkeypatch.setenv("CC_WW_MINCHUNKS", "2")
    parity.check_convs("cpu", cases=parity.CONV_CASES_WINO, tcases=[])
    parity.check_conv_groups("cpu", cases=((2, 40, 5, 8, 136, 16, 1), (1, 32, 4, 16, 72, 8, 1)))        # G = 3 problems per launch


def test_convs_winograd_weight_gradient_padded_rows(monkeypatch):
    # widths that are not multiples of 4: zero-padded copies of dY / x, then the same kernel (conv.hip k_pad_rows); single and grouped
    for k in ("CC_WWP_MINQ", "CC_WWP_MINM", "CC_WWP_MINC", "CC_WW_MINM", "CC_WW_MINC"):
        monkeypatch.setenv(k, "1")
    monkeypatch.setenv("CC_WW_MINCHUNKS", "2")
    parity.check_convs("cpu", cases=parity.CONV_CASES_WINO_PADW_SMALL, tcases=[])
    parity.check_conv_groups("cpu", cases=((2, 12, 5, 10, 40, 16, 1),))


def test_convs_winograd_padded_input(monkeypatch):
    # widths that are not multiples of 4: forward and data-gradient on the Winograd kernel over a zero-padded copy of the input
    for k in ("CC_WINOP_MINM", "CC_WINOP_MINC", "CC_WINOP_MINQ", "CC_WINO_MINM", "CC_WINO_MINC", "CC_WINO_MINQ"):
        monkeypatch.setenv(k, "1")
    monkeypatch.setenv("CC_WINO_TRACE", "1")
    parity.check_convs("cpu", cases=parity.CONV_CASES_WINO_PADIN_SMALL, tcases=[], prepack=True)
    # the same for G = 3 parallel branches in one launch (one copy launch pads all three inputs)
    parity.check_conv_groups("cpu", cases=((2, 12, 5, 26, 40, 16, 1), (1, 16, 4, 13, 24, 24, 1)))


def test_weight_gradient_list():
    parity.check_wgrad_list("cpu")


def test_weight_gradient_list_thin():
    # thin layers inside a list share launches per kernel instance (k_wgrad_thin_multi)
    parity.check_wgrad_list("cpu", shapes=parity.WGRAD_LIST_SHAPES_THIN, groups={3: 3})


def test_weight_gradient_list_winograd(monkeypatch):
    for k in ("CC_WW_MINQ", "CC_WW_MINM", "CC_WW_MINC", "CC_WWP_MINQ", "CC_WWP_MINM", "CC_WWP_MINC"):
        monkeypatch.setenv(k, "1")
    monkeypatch.setenv("CC_WW_MINCHUNKS", "2")
    parity.check_wgrad_list("cpu", shapes=parity.WGRAD_LIST_SHAPES_WINO)


def test_convs_thin_wgrad(monkeypatch):
    monkeypatch.setenv("CC_WGRAD_THIN_MINPIX", "0")      # route the small test maps through wgrad_thin.hip
    monkeypatch.setenv("CC_WGRAD_THIN_UPB", "8")
    parity.check_convs("cpu", cases=parity.CONV_CASES_THIN)


def test_convs_head_kernels(monkeypatch):
    monkeypatch.setenv("CC_HEAD_MINPIX", "1")            # the heads' forward pass on k_conv_thinm (conv_heads.hip)
    monkeypatch.setenv("CC_HEAD_WGRAD_MINPIX", "1")      # ... and their weight gradients on k_wgrad_thinm
    parity.check_convs("cpu", cases=parity.CONV_CASES_HEADS, tcases=[])
    monkeypatch.setenv("CC_HEAD_WGRAD_WAVES", "1")       # strips of 8 rows (the plan of the large maps)
    monkeypatch.setenv("CC_HEAD_ROWS_MINPIX", "1")       # forward: 4 / 2 rows per work-item (the plan of the large maps)
    parity.check_convs("cpu", cases=parity.CONV_CASES_HEADS, tcases=[])


def test_head_gradient_accumulators_match_the_engine_sums():
    # several loss terms on the same network outputs: with LF.head_grads active (the trainer's step) every term scales its gradients
    # into the per-tensor accumulator (cc_scale_acc_jobs) and ONE term hands it to autograd; the result must be what the autograd
    # engine sums up from the terms' separate tensors
    import torch
    from cc_amd import loss_functions as LF
    g = torch.Generator().manual_seed(5)
    img = torch.rand(2, 3, 24, 40, generator=g) * 2 - 1
    masks = [torch.sigmoid(torch.randn(2, 4, 24 >> s, 40 >> s, generator=g)).requires_grad_(True) for s in range(3)]

    def total():
        LF.pyramid_cache.clear()
        return 0.3 * LF.explainability_loss(masks) + 0.7 * LF.smooth_loss(masks) + 0.2 * LF.edge_aware_smoothness_loss(img, masks) \
            + 0.1 * LF.smooth_loss([masks[1], masks[1]])          # (the same tensor at two positions of one term)

    want = torch.autograd.grad(total(), masks)
    # repeated: the duplicate position used to share ONE launch with the first one ('=' and '+=' of unordered jobs: a race that lost
    # a contribution in most, not all, runs); now it goes into a launch of its own
    for _ in range(12):
        LF.head_grads.begin()
        try:
            assert LF.head_grads.active
            got = torch.autograd.grad(total(), masks)
        finally:
            LF.head_grads.end()
        for a, b in zip(got, want):
            assert float((a - b).abs().max()) <= 1e-6 * max(float(b.abs().max()), 1e-30), float((a - b).abs().max())
    # a second backward through the same term would add into the accumulators twice: refused, not silently doubled
    LF.head_grads.begin()
    try:
        t = total()
        torch.autograd.grad(t, masks, retain_graph=True)
        import pytest
        with pytest.raises(RuntimeError):
            torch.autograd.grad(t, masks)
    finally:
        LF.head_grads.end()


def test_cost_volume():
    parity.check_corr("cpu")
    parity.check_corr_patch("cpu")


def test_batch_norm():
    parity.check_batch_norm("cpu")


def test_upsample2x():
    parity.check_upsample2x("cpu")


def test_concat_gradient_slices():
    parity.check_concat_gradient_slices("cpu")


def test_adam_and_segments():
    parity.check_adam("cpu")


def test_conv_groups():
    parity.check_conv_groups("cpu")


def test_conv_launch_list():
    parity.check_conv_list("cpu")


def test_feature_warp_deterministic_scatter():
    parity.check_feature_warp_deterministic("cpu")


def test_pixel2cam_cam2pixel_gradients():
    parity.check_pixel2cam_cam2pixel_grads("cpu")


def test_bias_gradient_table():
    parity.check_bias_grad_table("cpu")


def test_sum_strided():
    parity.check_sum_strided("cpu")
