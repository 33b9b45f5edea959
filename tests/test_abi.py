"""The C-ABI library loads and exports every symbol include/ccengine.h declares (no compute: no GPU here);
the product refuses CPU tensors and a missing library loudly (no CPU fallback)."""
import ctypes
import os

import pytest
import torch

from cc_amd import _lib, build


def test_every_declared_symbol_is_exported():
    lib = build.build()
    sigs = _lib.parse_header()
    assert len(sigs) >= 35
    dll = ctypes.CDLL(lib)
    for name in sigs:
        assert hasattr(dll, name), name
    for need in ("cc_inverse_warp_fwd", "cc_inverse_warp_bwd", "cc_pose2flow_fwd", "cc_flow_warp_fwd", "cc_feature_warp_bwd",
                 "cc_ssim_photo_fwd", "cc_ssim_photo_bwd", "cc_consensus_target", "cc_pyramid_build", "cc_smooth2_fwd_bwd",
                 "cc_bce_ones_fwd_bwd", "cc_corr9x9_fwd", "cc_corr9x9_bwd", "cc_conv2d_fwd", "cc_conv2d_dgrad",
                 "cc_conv2d_wgrad", "cc_adam_step", "cc_version"):
        assert need in sigs, need


def test_library_contains_gfx950_code_objects():
    lib = build.build()
    blob = open(lib, "rb").read()
    assert b"gfx950" in blob and b"k_gather_gemm" in blob and b"k_ssim_tile" in blob


def test_no_cpu_fallback():
    build.build()
    e = _lib.Engine()                 # loads the real HIP library
    assert e.require_device
    x = torch.zeros(1, 3, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU path"):
        e.call("cc_ssim_fwd", x, x, x, 0, 1, 8, 8, None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.Engine(path=os.path.join(os.path.dirname(_lib.LIB_PATH), "does_not_exist.so"))


def test_product_never_imports_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "cc_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_product_library_reads_no_environment_and_keeps_no_timing_state():
    """the product library is stateless -- kernel-selection switches and the per-kernel timing registry exist in
    the tools build only (tools/_bin/libccengine_tools.so, -DCC_TOOLS)."""
    import subprocess
    lib = build.build()
    syms = subprocess.run(["nm", "-D", lib], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in syms
    dll = ctypes.CDLL(lib)
    assert dll.cc_is_tools_build() == 0
    assert dll.cc_timing_enable(1) != 0 and dll.cc_timing_collect(None, 0) == 0
    tools = build.build_tools()
    assert "getenv" in subprocess.run(["nm", "-D", tools], capture_output=True, text=True, check=True).stdout
    assert ctypes.CDLL(tools).cc_is_tools_build() == 1


def test_product_package_reads_no_environment():
    """The host glue takes its A/B switches from cc_amd.config.debug (set explicitly by tools/ab_env.py); the only environment
    variable the package looks at is HIPCC, in the build helper."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "cc_amd")):
        for f in files:
            if f.endswith(".py") and f != "build.py":
                src = open(os.path.join(dirpath, f)).read()
                assert "os.environ" not in src and "getenv" not in src, f
    from cc_amd import config
    from tools import ab_env
    saved = dict(vars(config.debug))
    try:
        got = ab_env.apply({"CC_NO_SUM_N": "1", "CC_FORCE_COMM": "1", "CC_CAPTURE_MODE": "relaxed", "CC_NO_WGRAD_LIST": "0"})
        assert got == {"no_sum_n": True, "force_comm": True, "capture_mode": "relaxed"}
        assert config.debug.no_sum_n and config.debug.force_comm and not config.debug.no_wgrad_list
    finally:
        for k in list(vars(config.debug)):
            delattr(config.debug, k)
        for k, v in saved.items():
            setattr(config.debug, k, v)
