#!/bin/bash
# kernel parity tests + same-box A/B of environment switches: bash tools/gpu_r2j.sh TAG VAR=VAL ...
TAG=${1:-r02j}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_nets_gpu.py -m gpu -q -x ) > gpurun_out/pytest_k_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|^E  |Error" gpurun_out/pytest_k_$TAG.log | tail -12
bash tools/gpu_ab_env.sh $TAG "$@"
