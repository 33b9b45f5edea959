#!/bin/bash
# side-stream weight gradients: correctness (step / headline tests with the switch on) + same-box A/B
TAG=${1:-r02i}
mkdir -p gpurun_out
export TMPDIR=/tmp
( CC_WGRAD_SIDE_STREAM=1 timeout 600 python -m pytest tests/test_headline_gpu.py tests/test_nets_gpu.py -m gpu -q -x ) > gpurun_out/pytest_side_$TAG.log 2>&1; echo "pytest(side) rc=$?"
grep -E "passed|failed|FAILED|^E  |Error" gpurun_out/pytest_side_$TAG.log | tail -12
bash tools/gpu_ab_env.sh $TAG CC_WGRAD_SIDE_STREAM=1
