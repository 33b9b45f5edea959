"""Build libccengine.so (hand-written HIP kernels + the C ABI) for gfx950 with hipcc.

In-tree artefact: cc_amd/libccengine.so (git-ignored, travels to the GPU box with
the gpurun snapshot).  hipcc cross-compiles without a GPU.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libccengine.so")
OBJ = os.path.join(HERE, "csrc", "_obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: FMAs only where the source says fmaf (SURVEY.md appendix D bit-exact recipe)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wno-unused-result", "-I", os.path.join(os.path.dirname(HERE), "include")]


# per-file flags.  ssim.hip: no SLP vectorisation -- it packs the separable filters' FMAs into v_pk_fma_f32 / v_pk_add_f32, which cost
# more on gfx950 than the scalar instructions they replace (profiles/r04_ab_round4.txt: the packed forward filter ran 47 % slower),
# and narrows half-used 16-byte LDS reads into ds_read2_b32 pairs
FILE_FLAGS = {"ssim.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def source_hash():
    """First 32 bits of the SHA-1 over the kernel sources + headers (-> cc_version())."""
    import hashlib
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) +
                   glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return int(h.hexdigest()[:8], 16)


TOOLS_OUT = os.path.join(os.path.dirname(HERE), "tools", "_bin", "libccengine_tools.so")


def build_tools(verbose=False, force=False):
    """The same sources with -DCC_TOOLS: kernel-selection switches (environment variables) and the per-kernel timing registry
    compiled in -> tools/_bin/libccengine_tools.so.  Loaded by bench.py for its instrumented eager step and by the A/B scripts
    under tools/ (CC_LIB_PATH); never by the product path."""
    return build(verbose, force, out=TOOLS_OUT, obj=os.path.join(CSRC, "_obj_tools"), extra=["-DCC_TOOLS"])


def build(verbose=False, force=False, out=None, obj=None, extra=()):
    OUT_, OBJ_ = out or OUT, obj or OBJ
    return _build(verbose, force, OUT_, OBJ_, list(extra))


def _build(verbose, force, OUT, OBJ, extra):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h"))
    jobs = []
    objs = []
    sh = source_hash()
    stamp = os.path.join(OBJ, "version.hash")
    hash_changed = not os.path.exists(stamp) or open(stamp).read().strip() != str(sh)
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        is_ver = os.path.basename(src) == "version.hip"
        if force or _stale(obj, [src] + headers) or (is_ver and hash_changed):
            jobs.append([HIPCC] + FLAGS + extra + FILE_FLAGS.get(os.path.basename(src), []) +
                        (["-DCC_SRC_HASH=%du" % sh] if is_ver else []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        return r

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(OUT, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    with open(stamp, "w") as f:
        f.write(str(sh))
    return OUT


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
    if "--tools" in sys.argv:
        print(build_tools(verbose=True, force="--force" in sys.argv))
