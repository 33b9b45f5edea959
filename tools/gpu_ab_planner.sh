#!/bin/bash
# several env configurations of the conv planner on the same box
mkdir -p gpurun_out
export TMPDIR=/tmp
# the kernel-selection switches exist in the tools build of the library only (cc_amd/build.py build_tools)
export CC_LIB_PATH=${CC_LIB_PATH:-$PWD/tools/_bin/libccengine_tools.so}
run() { ( env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing ) > gpurun_out/bench_x.log 2>&1; echo "$* : $(grep timed gpurun_out/bench_x.log)"; }
run CC_X=0
run CC_CONV_SPLIT_TARGET=256
run CC_CONV_SPLIT_TARGET=384
run CC_CONV_BM64_BELOW=256
run CC_CONV_BM64_BELOW=256 CC_CONV_SPLIT_TARGET=256
run CC_CONV_BM64_BELOW=512
run CC_X=0
