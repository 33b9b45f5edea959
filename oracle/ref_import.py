"""TEST INFRASTRUCTURE ONLY -- loader for the *unmodified* reference modules.

Usable where ``/root/reference`` exists (the build container) or where the
archive ``oracle/_ref/ccref.zip`` has been built from it (``make -C oracle``:
the nine unmodified files of SURVEY.md 8a, git-ignored, shipped to the GPU box
like a built library).  It is used by ``oracle/make_golden.py`` to generate the
fixtures under ``tests/golden/`` and by ``bench.py``'s ``cpu_baseline`` child
(kind "reference").  Nothing in ``cc_amd/`` may import this file.

Two shims are needed to run the reference on CPU:

* a module called ``spatial_correlation_sampler`` (third-party CUDA extension,
  PyPI ``spatial-correlation-sampler``, *unpinned* in requirements.txt:13 and
  not vendored) -- we restate its published semantic from the call sites
  models/back2future.py:15-25: ``out[b,i,j,y,x] = sum_c in1[b,c,y,x] *
  in2[b,c,y+(i-r)*dp, x+(j-r)*dp]`` with zero padding.  "parity unpinned": the
  reference holds no test of this op.
* ``torch.Tensor.cuda`` / ``Module.cuda`` made a no-op when no GPU is present
  (models/back2future.py:58-59,301-302,311 call ``.cuda()`` unconditionally).
"""
import importlib
import importlib.util
import os
import sys
import types
import warnings

import torch

REF_ROOT = os.environ.get("CC_REFERENCE_ROOT", "/root/reference")
REF_ZIP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "ccref.zip")


def _unpack_archive():
    """The reference tree is absent (GPU box) but oracle/_ref/ccref.zip travelled: unpack it to a temporary directory."""
    global REF_ROOT
    import atexit
    import shutil
    import tempfile
    import zipfile
    d = tempfile.mkdtemp(prefix="ccref_")
    atexit.register(shutil.rmtree, d, ignore_errors=True)
    with zipfile.ZipFile(REF_ZIP) as z:
        z.extractall(d)
    REF_ROOT = d


def reference_available():
    if os.path.isfile(os.path.join(REF_ROOT, "inverse_warp.py")):
        return True
    if os.path.isfile(REF_ZIP):
        _unpack_archive()
        return os.path.isfile(os.path.join(REF_ROOT, "inverse_warp.py"))
    return False


def _corr_sample(input1, input2, kernel_size=1, patch_size=1, stride=1,
                 padding=0, dilation=1, dilation_patch=1):
    assert kernel_size == 1 and stride == 1 and padding == 0 and dilation == 1
    from . import corr as _corr
    return _corr.correlation_volume(input1, input2, patch_size, dilation_patch)


def install_shims():
    if "spatial_correlation_sampler" not in sys.modules:
        m = types.ModuleType("spatial_correlation_sampler")
        m.spatial_correlation_sample = _corr_sample
        sys.modules["spatial_correlation_sampler"] = m
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self


_cache = {}


def load(align_corners=None):
    """Return a namespace with the reference's inverse_warp / loss_functions /
    ssim modules and the six model classes used by BASELINE.json's configs.

    align_corners: None -> run the reference exactly as it executes under this
    torch (grid_sample default, i.e. False + a warning); True/False -> patch
    ``F.grid_sample``'s default for the duration of the calls
    made through the returned namespace (the patch stays installed).
    """
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    install_shims()
    warnings.filterwarnings("ignore")
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    ns = types.SimpleNamespace()
    ns.inverse_warp = importlib.import_module("inverse_warp")
    ns.ssim = importlib.import_module("ssim")
    ns.loss_functions = importlib.import_module("loss_functions")
    for name, fname in [("DispResNet6", "DispResNet6"), ("PoseNetB6", "PoseNetB6"),
                        ("MaskNet6", "MaskNet6"), ("DispNetS", "DispNetS"),
                        ("PoseExpNet", "PoseExpNet"), ("back2future", "back2future")]:
        key = "ccref_models_" + fname
        if key not in _cache:
            spec = importlib.util.spec_from_file_location(
                key, os.path.join(REF_ROOT, "models", fname + ".py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            _cache[key] = mod
        setattr(ns, name, _cache[key])
    set_align_corners(align_corners)
    return ns


_orig_grid_sample = torch.nn.functional.grid_sample


def set_align_corners(ac):
    F = torch.nn.functional
    if ac is None:
        F.grid_sample = _orig_grid_sample
        return

    def patched(input, grid, mode="bilinear", padding_mode="zeros", align_corners=None):
        return _orig_grid_sample(input, grid, mode=mode, padding_mode=padding_mode,
                                 align_corners=ac if align_corners is None else align_corners)
    F.grid_sample = patched
