#!/bin/bash
# round 4, first contact of the Winograd kernel with the hardware: parity (per call + prepacked, split-K + fused), then the
# per-shape table against the direct kernel
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "convs" > gpurun_out/r4a_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4a_pytest.log
tail -5 gpurun_out/r4a_pytest.log
timeout 900 python tools/wino_bench.py --iters 5 > gpurun_out/r4a_wino_bench.txt 2> gpurun_out/r4a_wino_bench.err
tail -50 gpurun_out/r4a_wino_bench.txt
tail -5 gpurun_out/r4a_wino_bench.err
