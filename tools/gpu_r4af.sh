#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "winograd or wgrad" 2>&1 | tail -3
bash tools/gpu_ab_env.sh r4af CC_NO_WINO_WGRAD_PAD=1 "CC_WWP_MINM=64 CC_WWP_MINC=64" "CC_WWP_MINQ=16" "CC_WWP_MINM=128 CC_WWP_MINC=128"
