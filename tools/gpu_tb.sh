#!/bin/bash
# GPU parity tests + one bench line (no CPU baseline)
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|^E  " gpurun_out/pytest_gpu.log | tail -12
( timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing ) > gpurun_out/bench_graph.log 2>&1; echo "bench rc=$?"
grep -E "bench\]" gpurun_out/bench_graph.log | tail -2
