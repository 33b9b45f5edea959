#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r4c_probe.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "convs" 2>&1 | tail -3
for A in 0 1 23 32; do
  CC_WINO_ABL=$A timeout 200 python tools/wino_probe.py 2>/dev/null | tail -1 >> gpurun_out/r4c_probe.txt
done
PROBE_M=128 CC_WINO_ABL=0 timeout 200 python tools/wino_probe.py 2>/dev/null | tail -1 >> gpurun_out/r4c_probe.txt
cat gpurun_out/r4c_probe.txt
timeout 600 python tools/wino_bench.py --iters 5 2>/dev/null | grep -v kernels > gpurun_out/r4c_wino_bench.txt
cat gpurun_out/r4c_wino_bench.txt
