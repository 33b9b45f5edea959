#!/bin/bash
# A/B of one env switch inside the same GPU box: bash tools/gpu_ab.sh VAR
mkdir -p gpurun_out
export TMPDIR=/tmp
# the kernel-selection switches exist in the tools build of the library only (cc_amd/build.py build_tools)
export CC_LIB_PATH=${CC_LIB_PATH:-$PWD/tools/_bin/libccengine_tools.so}
V=${1:-CC_NO_CLASS_MERGE}
for rep in 1 2; do
( timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing ) > gpurun_out/bench_a.log 2>&1; echo "default   : $(grep timed gpurun_out/bench_a.log)"
( env $V=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing ) > gpurun_out/bench_b.log 2>&1; echo "$V=1 : $(grep timed gpurun_out/bench_b.log)"
done
