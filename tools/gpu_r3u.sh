#!/bin/bash
# the other benchmark configurations on the final build (one box): default, freeze variant, config 2, C5 shape, per-GPU batch 8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r3u
( timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 ) > gpurun_out/bench_${T}_default.log 2> gpurun_out/bench_${T}_default.err; echo "default: $(grep timed gpurun_out/bench_${T}_default.err)"
( timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --freeze ) > gpurun_out/bench_${T}_freeze.log 2> gpurun_out/bench_${T}_freeze.err; echo "freeze: $(grep timed gpurun_out/bench_${T}_freeze.err)"
( timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --config c2 ) > gpurun_out/bench_${T}_c2.log 2> gpurun_out/bench_${T}_c2.err; echo "c2: $(grep timed gpurun_out/bench_${T}_c2.err)"
( timeout 400 python bench.py --no-cpu-baseline --no-kernel-timing --height 512 --width 1664 --batch 2 ) > gpurun_out/bench_${T}_c5.log 2> gpurun_out/bench_${T}_c5.err; echo "c5 shape: $(grep timed gpurun_out/bench_${T}_c5.err)"
( timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --batch 8 ) > gpurun_out/bench_${T}_b8.log 2> gpurun_out/bench_${T}_b8.err; echo "b8: $(grep timed gpurun_out/bench_${T}_b8.err)"
