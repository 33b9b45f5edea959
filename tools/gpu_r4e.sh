#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r4e}
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "convs or conv_groups" 2>&1 | tail -2
rm -f gpurun_out/${TAG}_probe.txt
for M in 64 256; do
  PROBE_M=$M CC_WINO_ABL=0 timeout 200 python tools/wino_probe.py 2>/dev/null | tail -1 >> gpurun_out/${TAG}_probe.txt
done
cat gpurun_out/${TAG}_probe.txt
timeout 600 python tools/wino_bench.py --iters 5 2>/dev/null | grep -v kernels > gpurun_out/${TAG}_wino_bench.txt
head -30 gpurun_out/${TAG}_wino_bench.txt
( timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing ) > gpurun_out/${TAG}_bench.log 2>&1; grep -E "bench\]" gpurun_out/${TAG}_bench.log | tail -1
