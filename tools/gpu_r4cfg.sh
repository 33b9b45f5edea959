#!/bin/bash
# the other benchmark configurations on the current build (no tests): freeze variant, config 2, config 5 per-GPU shape, per-GPU batch 8
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing ) > gpurun_out/bench_cfg_full.log 2> gpurun_out/bench_cfg_full.err; echo "full: $(grep timed gpurun_out/bench_cfg_full.err)"
( timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --freeze ) > gpurun_out/bench_cfg_freeze.log 2> gpurun_out/bench_cfg_freeze.err; echo "freeze: $(grep timed gpurun_out/bench_cfg_freeze.err)"
( timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --config c2 ) > gpurun_out/bench_cfg_c2.log 2> gpurun_out/bench_cfg_c2.err; echo "c2: $(grep timed gpurun_out/bench_cfg_c2.err)"
( timeout 400 python bench.py --no-cpu-baseline --no-kernel-timing --height 512 --width 1664 --batch 2 ) > gpurun_out/bench_cfg_c5.log 2> gpurun_out/bench_cfg_c5.err; echo "c5 shape: $(grep timed gpurun_out/bench_cfg_c5.err)"
( timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --batch 8 ) > gpurun_out/bench_cfg_b8.log 2> gpurun_out/bench_cfg_b8.err; echo "b8: $(grep timed gpurun_out/bench_cfg_b8.err)"
