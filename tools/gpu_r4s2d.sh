#!/bin/bash
# round 4 (second session): conv_heads.hip k_conv_thinm (forward of the prediction heads on the large maps): parity subset + same-box A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_nets_gpu.py -q -x -k "convs or net_forward or step" > gpurun_out/pytest_r4s2d.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_r4s2d.log
bash tools/gpu_ab_env.sh r4s2d CC_NO_HEAD_KERNELS=2 CC_NO_HEAD_KERNELS=0 CC_NO_HEAD_KERNELS=2
