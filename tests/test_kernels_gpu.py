"""Parity tests proper: the real libccengine.so (gfx950) through the C ABI vs the oracle (CPU) on the same
seeded inputs.  Needs an MI355X: run with `-m gpu` via gpurun."""
import pytest
import torch

import parity

pytestmark = pytest.mark.gpu


def test_library_is_the_hip_build():
    from cc_amd import _lib
    e = _lib.engine()
    assert e.path.endswith("cc_amd/libccengine.so") and e.require_device
    assert torch.cuda.is_available()


def test_library_is_built_from_the_sources_in_the_tree():
    """The .so travels prebuilt with the snapshot (it is git-ignored): a stale one must fail loudly -- cc_version() carries the
    hash of the kernel sources + headers it was compiled from (cc_amd/build.py source_hash, csrc/version.hip)."""
    from cc_amd import _lib, build
    v = int(_lib.engine().fn["cc_version"]())
    assert v & 0xFFFFFFFF == build.source_hash(), "cc_amd/libccengine.so was built from other sources: run python -m cc_amd.build"
    assert v >> 32 == 2
    assert _lib.engine().fn["cc_is_tools_build"]() == 0


def test_warps():
    parity.check_warps("cuda")


def test_warps_full_size():
    # sampling coordinates reach ~800 px here: a last-ulp difference in P (device sin/cos vs host) is ~1e-4 px
    parity.check_warps("cuda", B=2, H=256, W=832, smooth=3, atol=3e-4)


def test_ssim():
    parity.check_ssim("cuda")
    parity.check_ssim("cuda", cases=((2, 256, 832, 2),))


def test_losses():
    # low-pass frames: the oracle runs on THIS host CPU, whose P = K.[R|t] differs in the last ulp from the device's
    parity.check_losses("cuda", smooth=3)


def test_occluded_flow_loss():
    parity.check_occluded_photo_loss("cuda")


def test_pyramid():
    parity.check_pyramid("cuda")


def test_warps_bit_exact_vs_reference_golden(golden_dir):
    r = parity.check_warps_bit_exact_vs_golden("cuda", golden_dir)
    print(r)
    for tag, d in r.items():
        assert d["tap_flip_rate_device_P"] <= 2e-3, (tag, d)      # measured: see the printed dict
        for k in ("inverse_warp", "pose2flow", "flow_warp", "feature_warp", "grid"):
            assert d[k] == 1.0, (tag, k, d)


def test_losses_vs_reference_golden(golden_dir):
    print(parity.check_losses_vs_golden("cuda", golden_dir))
    print("measured against the bars:", parity.MEASURED)


def test_convs():
    parity.check_convs("cuda")


def test_convs_winograd():
    # 3x3 / stride-1 layers on the Winograd F(2x2, 3x3) kernel: split-K cases (small maps), then launches whose workgroups hold the
    # whole reduction (fused epilogue), each per call and through the per-step weight images
    parity.check_convs("cuda", cases=parity.CONV_CASES_WINO, tcases=[], prepack=True)
    parity.check_convs("cuda", cases=parity.CONV_CASES_WINO_LARGE, tcases=[], prepack=True)
    # (the large cases also run their weight gradients on the Winograd F(3x3, 2x2) kernel, wino_wgrad.hip; grouped form below)
    parity.check_conv_groups("cuda", cases=((2, 48, 16, 32, 64, 48, 1),))
    # widths that are not multiples of 4 (the 8x26 / 4x13 levels): weight gradients over zero-padded copies (conv.hip k_pad_rows)
    parity.check_convs("cuda", cases=parity.CONV_CASES_WINO_PADW, tcases=[])
    parity.check_conv_groups("cuda", cases=((4, 96, 8, 26, 128, 96, 1),))


def test_convs_winograd_nonsmooth_epilogues_at_product_sizes():
    # the PRODUCT library at sizes the planner does not slice: ReLU / LeakyReLU + residual in the Winograd forward epilogue, the
    # deferred activation backward in the data-gradient epilogue (EPI_GRAD), on the 64 x 64 and both 32 x 32 instances; activation
    # derivative pinned to the device's mask, flips counted and shown to sit at |pre-activation| < 1e-5 (parity.check_convs_act_pinned)
    from cc_amd import _lib
    assert _lib.engine().fn["cc_is_tools_build"]() == 0
    rep = parity.check_convs_act_pinned("cuda")
    for case, nflip, err in rep:
        print("pinned-activation case %s: %d derivative flips, worst error %.2e" % (case, nflip, err))


def test_convs_winograd_padded_input(monkeypatch):
    # 8x26-style maps: forward / data-gradient on the Winograd kernel over a zero-padded copy of the input (product thresholds),
    # then small shapes steered there through the tools build
    from cc_amd import _lib, build
    parity.check_convs("cuda", cases=parity.CONV_CASES_WINO_PADIN, tcases=[], prepack=True)
    for k in ("CC_WINOP_MINM", "CC_WINOP_MINC", "CC_WINOP_MINQ", "CC_WINO_MINM", "CC_WINO_MINC", "CC_WINO_MINQ"):
        monkeypatch.setenv(k, "1")
    with _lib.use_library(build.build_tools()) as e:
        assert e.fn["cc_is_tools_build"]() == 1
        parity.check_convs("cuda", cases=parity.CONV_CASES_WINO_PADIN_SMALL, tcases=[], prepack=True)
        parity.check_conv_groups("cuda", cases=((2, 12, 5, 26, 40, 16, 1), (1, 16, 4, 13, 24, 24, 1)))      # G = 3 branches per launch


def test_weight_gradient_list():
    # the end-of-stage flush of the weight-gradient queue: groups of different shapes in one cc_conv2d_wgrad_list call
    parity.check_wgrad_list("cuda")


def test_weight_gradient_list_thin():
    # ... with thin layers in the list: the problems of one kernel instance share launches (k_wgrad_thin_multi)
    parity.check_wgrad_list("cuda", shapes=parity.WGRAD_LIST_SHAPES_THIN, groups={3: 3})


def test_weight_gradient_list_winograd(monkeypatch):
    # ... with 3x3 / stride-1 layers on the Winograd weight-gradient kernel in the list: they share multi-geometry launches
    # (k_wino_wgrad_multi); small shapes steered there through the tools build's switches, product-size shapes at product thresholds
    from cc_amd import _lib, build
    for k in ("CC_WW_MINQ", "CC_WW_MINM", "CC_WW_MINC", "CC_WWP_MINQ", "CC_WWP_MINM", "CC_WWP_MINC"):
        monkeypatch.setenv(k, "1")
    monkeypatch.setenv("CC_WW_MINCHUNKS", "2")
    with _lib.use_library(build.build_tools()) as e:
        assert e.fn["cc_is_tools_build"]() == 1
        parity.check_wgrad_list("cuda", shapes=parity.WGRAD_LIST_SHAPES_WINO)
    for k in ("CC_WW_MINQ", "CC_WW_MINM", "CC_WW_MINC", "CC_WWP_MINQ", "CC_WWP_MINM", "CC_WWP_MINC", "CC_WW_MINCHUNKS"):
        monkeypatch.delenv(k)
    # the PRODUCT library at its own thresholds: k_wino_wgrad_multi with single problems, a G = 2 group (index 1) and a G = 3 group (index 3)
    assert _lib.engine().fn["cc_is_tools_build"]() == 0
    parity.check_wgrad_list("cuda", shapes=[(4, 128, 8, 28, 128, 3, 1, 1), (4, 96, 16, 52, 96, 3, 1, 1), (4, 128, 8, 26, 96, 3, 1, 1),
                                             (4, 64, 32, 104, 64, 3, 1, 1), (2, 12, 9, 14, 20, 3, 2, 1), (4, 192, 4, 16, 192, 3, 1, 1)],
                            groups={1: 2, 3: 3})


def test_convs_prepacked_weight_images():
    parity.check_convs("cuda", prepack=True)


def test_convs_thin_wgrad(monkeypatch):
    # the kernel-selection switches exist in the tools build only (cc_amd/build.py build_tools): small maps are steered through
    # wgrad_thin.hip there; the product library's own thresholds are covered by test_convs_full_size_thin_layers below
    from cc_amd import _lib, build
    monkeypatch.setenv("CC_WGRAD_THIN_MINPIX", "0")
    monkeypatch.setenv("CC_WGRAD_THIN_UPB", "8")
    with _lib.use_library(build.build_tools()) as e:
        assert e.fn["cc_is_tools_build"]() == 1
        parity.check_convs("cuda", cases=parity.CONV_CASES_THIN)
    assert _lib.engine().fn["cc_is_tools_build"]() == 0


def test_convs_head_kernels(monkeypatch):
    # small maps steered onto the heads' VALU forward kernel (tools build); the product thresholds: test_convs_full_size_thin_layers
    from cc_amd import _lib, build
    monkeypatch.setenv("CC_HEAD_MINPIX", "1")
    monkeypatch.setenv("CC_HEAD_WGRAD_MINPIX", "1")
    with _lib.use_library(build.build_tools()) as e:
        assert e.fn["cc_is_tools_build"]() == 1
        parity.check_convs("cuda", cases=parity.CONV_CASES_HEADS, tcases=[])
        monkeypatch.setenv("CC_HEAD_ROWS_MINPIX", "1")       # 4 / 2 rows per work-item on the small test maps too
        parity.check_convs("cuda", cases=parity.CONV_CASES_HEADS, tcases=[])


def test_convs_full_size_thin_layers():
    # the real thin layers of the 256x832 step (default thresholds): DispResNet6 iconv1 / head, MaskNet6 conv1, B2F feat1
    cases = [(2, 17, 256, 832, 16, 3, 1, 1, "relu", True, False), (2, 16, 256, 832, 1, 3, 1, 1, "sigmoid", True, False),
             (2, 15, 256, 832, 16, 7, 2, 3, "relu", True, False), (2, 3, 256, 832, 16, 3, 2, 1, "lrelu", True, False),
             (2, 32, 128, 416, 32, 7, 1, 3, "relu", True, False),
             # the 2-channel flow heads of Back2Future (k_conv_thinm<2> forward, k_wgrad_thinm<2> weight gradient, k_conv_thinc<2>
             # data-gradient of the layer behind them) and MaskNet6's 4-channel heads at their product sizes and thresholds
             (4, 32, 64, 208, 2, 3, 1, 1, None, True, False), (4, 2, 64, 208, 32, 3, 1, 1, "lrelu", True, False),
             (4, 32, 128, 416, 4, 3, 1, 1, "sigmoid", True, False)]
    from cc_amd import _lib
    assert _lib.engine().fn["cc_is_tools_build"]() == 0          # the product library, its own thresholds
    parity.check_convs("cuda", cases=cases, tcases=[(2, 48, 64, 208, 16, 4, 2, 1, 0, "relu")], tol=5e-5)


def test_cost_volume():
    parity.check_corr("cuda")
    parity.check_corr_patch("cuda")
    parity.check_corr("cuda", cases=((2, 32, 64, 208), (2, 196, 8, 26)))


def test_batch_norm():
    parity.check_batch_norm("cuda")
    parity.check_batch_norm("cuda", cases=((4, 16, 256, 832), (4, 64, 64, 208), (4, 128, 32, 104), (4, 512, 2, 7)))


def test_upsample2x():
    parity.check_upsample2x("cuda")
    parity.check_upsample2x("cuda", cases=((4, 2, 64, 208, 20.0), (4, 1, 128, 416, 1.0)))


def test_concat_gradient_slices():
    parity.check_concat_gradient_slices("cuda")


def test_adam_and_segments():
    parity.check_adam("cuda")


def test_conv_groups():
    parity.check_conv_groups("cuda")
    parity.check_conv_groups("cuda", cases=((4, 196, 16, 52, 128, 96, 1), (4, 64, 32, 104, 96, 32, 2)))


def test_conv_launch_list():
    parity.check_conv_list("cuda")


def test_feature_warp_deterministic_scatter():
    parity.check_feature_warp_deterministic("cuda")
    parity.check_feature_warp_deterministic("cuda", cases=((4, 32, 64, 208), (4, 128, 8, 26)))


def test_pixel2cam_cam2pixel_gradients():
    parity.check_pixel2cam_cam2pixel_grads("cuda")
    parity.check_pixel2cam_cam2pixel_grads("cuda", B=2, H=128, W=416)


def test_bias_gradient_table():
    parity.check_bias_grad_table("cuda")
    parity.check_bias_grad_table("cuda", cases=((4, 64, 64, 208, 0), (4, 16, 256, 832, 1), (4, 512, 2, 7, 0), (4, 32, 128, 416, 33)) * 9)


def test_sum_strided():
    parity.check_sum_strided("cuda")
    parity.check_sum_strided("cuda", cases=((4, 2, 64, 208, 4, 0), (4, 128, 32, 104, 3, 1), (4, 34, 16, 52, 8, 1)))
