#!/bin/bash
# same-box A/B of environment switches: bash tools/gpu_ab_env.sh TAG VAR=VAL [VAR=VAL ...]   (default run first and last)
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
# the kernel-selection switches exist in the tools build of the library only (cc_amd/build.py build_tools)
export CC_LIB_PATH=${CC_LIB_PATH:-$PWD/tools/_bin/libccengine_tools.so}
for V in default "$@" default; do
  if [ "$V" = default ]; then E=""; else E="$V"; fi
  F=$(echo "$V" | tr ' =/' '___')
  ( env $E timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 ) > gpurun_out/bench_${TAG}_$F.log 2> gpurun_out/bench_${TAG}_$F.err
  echo "$V: $(grep timed gpurun_out/bench_${TAG}_$F.err)"
done
