#!/bin/bash
# batched side-stream weight gradients: correctness with the switch on + same-box A/B over batch sizes
TAG=${1:-r02k}
mkdir -p gpurun_out
export TMPDIR=/tmp
( CC_WGRAD_SIDE_STREAM=1 timeout 600 python -m pytest tests/test_headline_gpu.py tests/test_nets_gpu.py -m gpu -q -x ) > gpurun_out/pytest_side_$TAG.log 2>&1; echo "pytest(side) rc=$?"
grep -E "passed|failed|FAILED|^E  |Error" gpurun_out/pytest_side_$TAG.log | tail -12
for V in default "CC_WGRAD_SIDE_STREAM=1 CC_WGRAD_SIDE_BATCH=8" "CC_WGRAD_SIDE_STREAM=1 CC_WGRAD_SIDE_BATCH=16" "CC_WGRAD_SIDE_STREAM=1 CC_WGRAD_SIDE_BATCH=40" default; do
  if [ "$V" = default ]; then E=""; else E="$V"; fi
  F=$(echo "$V" | tr ' =' '__')
  ( env $E timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 ) > gpurun_out/bench_${TAG}_$F.log 2> gpurun_out/bench_${TAG}_$F.err
  echo "$V: $(grep timed gpurun_out/bench_${TAG}_$F.err)"
done
