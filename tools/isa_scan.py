#!/usr/bin/env python
"""Find latency-chained loops in the gfx950 ISA of the kernel sources (no GPU needed).

  python tools/isa_scan.py [file.hip ...]        (default: every cc_amd/csrc/*.hip)

Compiles each file to assembly (hipcc -S --cuda-device-only) and reports, per kernel, every loop (backward-branch segment) that
contains vector loads AND `s_waitcnt vmcnt(0)`: (instructions, loads, vmcnt(0) waits, MFMAs).  A loop with one or two loads per
wait is a chain of load latencies -- the pattern that cost this engine 1.6 ms per step before the loads of several iterations were
issued together (NOTES.md, "Loads in flight")."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FILT = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"


def demangle(n):
    try:
        return subprocess.run([FILT, n], capture_output=True, text=True).stdout.strip()
    except OSError:
        return n


def scan(asm_path, max_loads_per_wait):
    lines = open(asm_path).read().split("\n")
    funcs, cur = {}, None
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        elif cur is not None:
            funcs[cur].append(ln)
        if ln.startswith(".Lfunc_end"):
            cur = None
    for fn, body in funcs.items():
        labels = {}
        for k, ln in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", ln)
            if m:
                labels[m.group(1)] = k
        hits = []
        for k, ln in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
            if m and m.group(1) in labels and labels[m.group(1)] < k:
                seg = body[labels[m.group(1)]:k]
                loads = sum(1 for x in seg if re.search(r"\b(global|flat|buffer)_load", x) and "lds" not in x)
                waits = sum(1 for x in seg if re.search(r"s_waitcnt.*vmcnt\(0\)", x))
                mfma = sum(1 for x in seg if "v_mfma" in x)
                if loads and waits and loads <= max_loads_per_wait * waits and len(seg) < 600:
                    hits.append((len(seg), loads, waits, mfma))
        if hits:
            print("%-100s %s" % (demangle(fn)[:100], sorted(set(hits))[:4]))


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "cc_amd", "csrc", "*.hip")))
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            out = os.path.join(tmp, os.path.basename(f) + ".s")
            r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I",
                                os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", out, f],
                               capture_output=True, text=True)
            if r.returncode != 0:
                print("%s: hipcc failed\n%s" % (f, r.stderr[-2000:]))
                continue
            print("== %s" % os.path.relpath(f, ROOT))
            scan(out, max_loads_per_wait=4)


if __name__ == "__main__":
    main()
