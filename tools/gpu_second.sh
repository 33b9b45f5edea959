#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -s ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
( timeout 300 python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline ) > gpurun_out/bench_nograph.log 2>&1; echo "bench nograph rc=$?"
( timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline ) > gpurun_out/bench_graph.log 2>&1; echo "bench graph rc=$?"
( timeout 400 python tools/conv_bench.py ) > gpurun_out/conv_bench.log 2>&1; echo "conv_bench rc=$?"
grep -E "passed|failed|FAILED|Error|acF|acT" gpurun_out/pytest_gpu.log | tail -20
tail -c 1500 gpurun_out/bench_nograph.log; echo; tail -c 1500 gpurun_out/bench_graph.log; echo; tail -25 gpurun_out/conv_bench.log
