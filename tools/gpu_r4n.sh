#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "wgrad or winograd or conv_groups" 2>&1 | tail -2
cd tools
timeout 600 python wino_wgrad_bench.py --iters 5 2>/dev/null | cut -c1-150 > ../gpurun_out/r4n_ww.txt
cd ..
cat gpurun_out/r4n_ww.txt
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r4n_bench.log 2>&1
grep -E "timed|^\{" gpurun_out/r4n_bench.log | cut -c1-200
