"""TEST INFRASTRUCTURE ONLY: compile the UNCHANGED cc_amd/csrc/*.hip sources for x86 against the
fiber-based HIP shim (tests/hipemu/shim) -> tests/hipemu/_build/libccengine_emu.so, so kernel
logic can be checked against the oracle without a GPU.  Never loaded by cc_amd itself."""
import glob
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "cc_amd", "csrc")
# CC_EMU_EXTRA="-DCC_LOADER_WAVE=1": a variant build of the kernel sources (A/B experiments) in its own directory
EXTRA = os.environ.get("CC_EMU_EXTRA", "").split()
BUILD = os.path.join(HERE, "_build" + ("_" + "".join(c if c.isalnum() else "_" for c in " ".join(EXTRA)) if EXTRA else ""))
OUT = os.path.join(BUILD, "libccengine_emu.so")
CXX = "/opt/rocm/lib/llvm/bin/clang++"
FLAGS = ["-x", "c++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fopenmp", "-mfma", "-mavx2",
         "-Wno-unknown-attributes", "-Wno-unused-value", "-DCC_TOOLS",      # tools switches: tests steer kernel selection with them
         "-I", os.path.join(HERE, "shim"), "-I", os.path.join(ROOT, "include")] + EXTRA


def build(only=None):
    os.makedirs(BUILD, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    if only:
        srcs = [s for s in srcs if os.path.basename(s)[:-4] in only]
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "shim", "hip", "*.h")) + \
        glob.glob(os.path.join(ROOT, "include", "*.h"))
    impl = os.path.join(BUILD, "impl.cpp")
    if not os.path.exists(impl):
        with open(impl, "w") as f:
            f.write("#define HIPEMU_IMPL\n#include <hip/hip_runtime.h>\n")
    jobs, objs = [], []
    for s in srcs + [impl]:
        o = os.path.join(BUILD, os.path.basename(s).rsplit(".", 1)[0] + ".o")
        objs.append(o)
        newest = max(os.path.getmtime(p) for p in [s] + hdrs)
        if not os.path.exists(o) or os.path.getmtime(o) < newest:
            jobs.append([CXX] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("emu build failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-6000:]))

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(OUT):
        run([CXX, "-shared", "-fPIC", "-fopenmp", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build())
