#!/bin/bash
# GPU tests (all) + A/B of the parked weight-gradient reductions
TAG=${1:-r02h}
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -x ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|^E  |Error" gpurun_out/pytest_gpu_$TAG.log | tail -12
bash tools/gpu_ab_env.sh $TAG CC_NO_WGRAD_DEFER=1
