#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_nets_gpu.py -q -x -k "weight_gradient_list or net_forward or checkpoint or step" 2>&1 | tail -3
bash tools/gpu_ab_env.sh r4ak CC_NO_WGRAD_LIST=1 CC_NO_WGRAD_LIST=0 CC_NO_WGRAD_LIST=1
