"""TEST INFRASTRUCTURE ONLY: bind cc_amd's Python glue to the x86 emulation build of the kernel
sources (tests/hipemu/build_emu.py) so the SAME autograd/C-ABI plumbing can be exercised on CPU
tensors.  Used by the `-m "not gpu"` tests; the GPU tests use the real libccengine.so."""
import contextlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import build_emu  # noqa: E402

from cc_amd import _lib  # noqa: E402


@contextlib.contextmanager
def emulated_engine(only=None):
    path = build_emu.build(only)
    prev = _lib._engine
    _lib._set_engine_for_tests(_lib.Engine(path, require_device=False))
    try:
        yield _lib._engine
    finally:
        _lib._set_engine_for_tests(prev)
