#!/bin/bash
# round 4 (second session): per-tensor gradient accumulators of the loss terms (cc_scale_acc_jobs) instead of the engine's pairwise adds
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_nets_gpu.py tests/test_headline_gpu.py tests/test_kernels_gpu.py -q -x -k "step or headline or losses or checkpoint" > gpurun_out/pytest_r4s2o.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_r4s2o.log
bash tools/gpu_ab_env.sh r4s2o CC_NO_HEAD_ACC=1 CC_NO_HEAD_ACC=0 CC_NO_HEAD_ACC=1
