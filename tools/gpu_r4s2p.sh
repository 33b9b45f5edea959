#!/bin/bash
# round 4 (second session): weight gradients of the prediction heads on k_wgrad_thinm (conv_heads.hip): parity subset, A/B, per-layer table
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_nets_gpu.py -q -x -k "head_kernels or full_size_thin or net_forward or step" > gpurun_out/pytest_r4s2p.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_r4s2p.log
bash tools/gpu_ab_env.sh r4s2p CC_NO_HEAD_KERNELS=4 CC_NO_HEAD_KERNELS=0 CC_NO_HEAD_KERNELS=4
bash tools/gpu_r4g.sh r4s2p > /dev/null 2>&1; grep -E "thinm" gpurun_out/layers_r4s2p.tsv; grep -E "thinm|wgrad_thin|reduce_table" gpurun_out/step_trace_r4s2p.txt
