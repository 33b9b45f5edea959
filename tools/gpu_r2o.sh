#!/bin/bash
# parity tests + A/B of env switches + one-step trace: bash tools/gpu_r2o.sh TAG VAR=VAL ...
TAG=${1:-r02o}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_nets_gpu.py tests/test_headline_gpu.py -m gpu -q -x ) > gpurun_out/pytest_k_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|^E  |Error" gpurun_out/pytest_k_$TAG.log | tail -12
bash tools/gpu_ab_env.sh $TAG "$@"
bash tools/gpu_prof.sh $TAG > gpurun_out/prof_$TAG.out 2>&1; grep -E "k_direct|one replayed" gpurun_out/step_trace_$TAG.txt
