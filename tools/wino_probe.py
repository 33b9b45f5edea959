#!/usr/bin/env python
"""Per-chunk time and per-workgroup fixed cost of the Winograd kernel: B=4, 64x256 maps, 64 output channels = exactly 256
workgroups (one per CU, one round), input channels swept -> duration = fixed + (Cin / 8) * t_chunk.  Forward only, kernel time
from the tools build's timing registry.  CC_WINO_ABL selects an ablation (tools build)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cc_amd import _lib, build, ops  # noqa: E402


def main():
    lib = build.TOOLS_OUT
    M = int(os.environ.get("PROBE_M", "64"))
    W = int(os.environ.get("PROBE_W", "256"))
    with _lib.use_library(lib) as eng:
        rows = []
        for Cin in (32, 64, 128, 256, 512):
            ops.packs.reset()
            x = torch.randn(4, Cin, 64, W, device="cuda")
            w = torch.randn(M, Cin, 3, 3, device="cuda") * 0.05
            b = torch.randn(M, device="cuda")
            ops.packs.recording = True
            with torch.no_grad():
                ops.conv2d(x, w, b, 1, 1, "lrelu")
                ops.packs.prepack_all()
                ops.conv2d(x, w, b, 1, 1, "lrelu")
                torch.cuda.synchronize()
                eng.call("cc_timing_enable", 1)
                for _ in range(10):
                    ops.conv2d(x, w, b, 1, 1, "lrelu")
                torch.cuda.synchronize()
            buf = ctypes.create_string_buffer(1 << 16)
            n = eng.fn["cc_timing_collect"](ctypes.addressof(buf), 1 << 16)
            for ln in buf.raw[:n].decode().splitlines():
                nm, cnt, ms, gf = ln.split("\t")
                rows.append((Cin, nm, float(ms) / int(cnt) * 1e3))
            ops.packs.invalidate()
        for r in rows:
            print("C %4d  %-22s %8.2f us" % r)
        us = {c: t for c, n, t in rows if n.startswith("k_wino")}
        if len(us) >= 2:
            cs = sorted(us)
            slope = (us[cs[-1]] - us[cs[0]]) / ((cs[-1] - cs[0]) / 8)
            print("ABL %s M %d W %d: per chunk %.3f us (= %.0f cycles at 2.4 GHz; MFMA floor 4096), fixed %.2f us" %
                  (os.environ.get("CC_WINO_ABL", "0"), M, W, slope, slope * 2400, us[cs[0]] - slope * cs[0] / 8))


if __name__ == "__main__":
    main()
