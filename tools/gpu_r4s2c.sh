#!/bin/bash
# round 4 (second session): conv_heads.hip k_conv_thinc (data-gradients of the prediction heads as VALU kernels): parity subset + same-box A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_nets_gpu.py -q -x -k "convs or net_forward or step" 2>&1 | tail -3
bash tools/gpu_ab_env.sh r4s2c CC_NO_HEAD_KERNELS=1 CC_NO_HEAD_KERNELS=0 CC_NO_HEAD_KERNELS=1
