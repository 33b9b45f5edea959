#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
( timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing ) > gpurun_out/bench_a.log 2>&1; echo "thread_local : $(grep timed gpurun_out/bench_a.log)"
( CC_CAPTURE_MODE=global timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing ) > gpurun_out/bench_b.log 2>&1; echo "global       : $(grep timed gpurun_out/bench_b.log)"
done
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -4
