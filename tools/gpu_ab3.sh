#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
( timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing ) > gpurun_out/bench_a.log 2>&1; echo "default     : $(grep timed gpurun_out/bench_a.log)"
( timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --elide-occ ) > gpurun_out/bench_b.log 2>&1; echo "--elide-occ : $(grep timed gpurun_out/bench_b.log)"
done
