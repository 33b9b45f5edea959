#!/bin/bash
# round 3: conv / net parity subset, then a same-box A/B of environment switches (tools build)
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -k "conv or net_forward or full_step" ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|^E  " gpurun_out/pytest_gpu_$TAG.log | tail -8
bash tools/gpu_ab_env.sh $TAG "$@"
