#!/bin/bash
# round 4: Winograd kernel parity + per-shape table + timing ablations of its main loop (tools build, CC_WINO_ABL)
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r4b}
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "convs" > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/${TAG}_pytest.log
tail -4 gpurun_out/${TAG}_pytest.log
timeout 900 python tools/wino_bench.py --iters 5 > gpurun_out/${TAG}_wino_bench.txt 2> gpurun_out/${TAG}_wino_bench.err
grep -v kernels gpurun_out/${TAG}_wino_bench.txt
for A in 1 2 4 7 8 16 23; do
  echo "== CC_WINO_ABL=$A" >> gpurun_out/${TAG}_abl.txt
  CC_WINO_ABL=$A timeout 300 python tools/wino_bench.py --iters 5 --quick 2>/dev/null | grep -v kernels >> gpurun_out/${TAG}_abl.txt
done
cat gpurun_out/${TAG}_abl.txt
