#!/bin/bash
# Where does k_conv_patch lose its time?  Ablation builds of the library (results are garbage, only the step time counts):
#   abl_dma: no LDS-DMA in the main loop (operands of the first stage are reused)   -> cost of DMA issue + waits
#   abl_lds: MFMA operands from registers instead of ds_read                        -> cost of the fragment reads
#   abl_bar: no __syncthreads() per stage                                           -> cost of the stage barrier
#   abl_store: every accumulator store of the BM >= 32 epilogues issued TWICE through a volatile access.  CONFOUNDED: the compiler
#              orders volatile accesses with waits, the +5 ms it showed is not the cost of the store tail (16-byte stores gained 0.22 ms)
#   abl_mfma: one VALU fma instead of each 32x32x2 MFMA of k_conv_patch (BM >= 32)  -> what the step costs without the matrix work
# Build here (no GPU needed):  bash tools/ablate_conv.sh build      Run on the GPU box:  bash tools/ablate_conv.sh run
set -e
if [ "$1" = build ]; then
  python - <<'PY'
from cc_amd import build
import os
root = os.path.dirname(os.path.dirname(build.HERE)) if False else os.path.dirname(build.HERE)
for tag, flag in (("abl_dma", "-DCC_ABLATE_DMA"), ("abl_lds", "-DCC_ABLATE_LDS"), ("abl_bar", "-DCC_ABLATE_BARRIER"), ("abl_mfma", "-DCC_ABLATE_MFMA"),
                  ("abl_mfma_dma", "-DCC_ABLATE_MFMA -DCC_ABLATE_DMA"), ("abl_mfma_lds", "-DCC_ABLATE_MFMA -DCC_ABLATE_LDS"),
                  ("abl_store", "-DCC_ABLATE_STORE"),
                  ("abl_all", "-DCC_ABLATE_MFMA -DCC_ABLATE_DMA -DCC_ABLATE_LDS -DCC_ABLATE_BARRIER")):
    if os.environ.get("ABL_ONLY") and tag not in os.environ["ABL_ONLY"].split():
        continue
    out = os.path.join(root, "tools", "_bin", "libccengine_%s.so" % tag)
    print(build.build(out=out, obj=os.path.join(build.CSRC, "_obj_" + tag), extra=["-DCC_TOOLS"] + flag.split()))
PY
  exit 0
fi
mkdir -p gpurun_out
export TMPDIR=/tmp
for L in ${ABL_LIST:-tools abl_dma abl_lds abl_bar abl_mfma tools}; do
  ( CC_LIB_PATH=$PWD/tools/_bin/libccengine_$L.so timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 3 ) > gpurun_out/bench_ablate_$L.log 2> gpurun_out/bench_ablate_$L.err || true
  echo "$L: $(grep timed gpurun_out/bench_ablate_$L.err)"
done
