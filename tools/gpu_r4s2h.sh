#!/bin/bash
# round 4 (second session): parked Winograd weight gradients in multi-geometry launches (k_wino_wgrad_multi): parity subset + same-box A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_nets_gpu.py -q -x -k "weight_gradient or head_kernels or net_forward or step" > gpurun_out/pytest_r4s2h.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_r4s2h.log
bash tools/gpu_ab_env.sh r4s2h CC_NO_WINO_WGRAD_LIST=1 CC_NO_WINO_WGRAD_LIST=0 CC_NO_WINO_WGRAD_LIST=1
