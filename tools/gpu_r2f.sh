#!/bin/bash
# tests + same-box A/B of the split-K / split-pixel planner targets
TAG=${1:-r02f}
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -x -s ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|^E  |Error|tap_flip|device kernels" gpurun_out/pytest_gpu_$TAG.log | tail -12
for V in default CC_WGRAD_SPLIT_TARGET=128 CC_WGRAD_SPLIT_TARGET=256 CC_CONV_SPLIT_TARGET=256 CC_CONV_SPLIT_TARGET=1024 CC_W3_SPLIT=128 CC_W3_SPLIT=512 default; do
  if [ $V = default ]; then E=""; else E="$V"; fi
  ( env $E timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 ) > gpurun_out/bench_${TAG}_$V.log 2> gpurun_out/bench_${TAG}_$V.err
  echo "$V: $(grep timed gpurun_out/bench_${TAG}_$V.err)"
done
