#!/bin/bash
# informational points beside the metric: per-GPU batch 8 and BASELINE config 2 (DispResNet6 + PoseNetB6 only)
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { ( timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing "$@" ) > gpurun_out/bench_x.log 2>&1; echo "$* : $(grep timed gpurun_out/bench_x.log)"; }
run --batch 4
run --batch 8
run --config c2
