#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r4f}
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "convs or conv_groups" 2>&1 | tail -4
cd tools && timeout 900 python wino_wgrad_bench.py --iters 5 2>/dev/null > ../gpurun_out/${TAG}_ww_bench.txt; cd ..
cat gpurun_out/${TAG}_ww_bench.txt
