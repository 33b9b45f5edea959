#!/bin/bash
# round 4 (second session): forward / data-gradient of the 8x26 512-channel layers on the Winograd kernel over a zero-padded input copy
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_nets_gpu.py -q -x -k "padded_input or test_convs or net_forward or step" > gpurun_out/pytest_r4s3c.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_r4s3c.log
bash tools/gpu_ab_env.sh r4s3c CC_NO_WINO_PAD=1 CC_NO_WINO_PAD=0 CC_NO_WINO_PAD=1
