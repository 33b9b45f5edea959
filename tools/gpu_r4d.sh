#!/bin/bash
# round 4: Winograd kernel (current build): conv parity, per-chunk probe, per-shape table, one bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r4d}
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "convs" 2>&1 | tail -3
rm -f gpurun_out/${TAG}_probe.txt
for A in 0 1 23 32; do
  CC_WINO_ABL=$A timeout 200 python tools/wino_probe.py 2>/dev/null | tail -1 >> gpurun_out/${TAG}_probe.txt
done
PROBE_M=128 CC_WINO_ABL=0 timeout 200 python tools/wino_probe.py 2>/dev/null | tail -1 >> gpurun_out/${TAG}_probe.txt
cat gpurun_out/${TAG}_probe.txt
timeout 600 python tools/wino_bench.py --iters 5 2>/dev/null | grep -v kernels > gpurun_out/${TAG}_wino_bench.txt
cat gpurun_out/${TAG}_wino_bench.txt
( timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing ) > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?"
grep -E "bench\]" gpurun_out/${TAG}_bench.log | tail -2
