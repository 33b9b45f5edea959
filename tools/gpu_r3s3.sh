#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/gpu_ab_env.sh r3s3 "CC_W3_MT_REM=1" "CC_W3_MT_REM=2"
CC_W3_MT_REM=1 CC_LIB_PATH=$PWD/tools/_bin/libccengine_tools.so timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "convs or groups" 2>&1 | tail -2
