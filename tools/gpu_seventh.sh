#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2 3; do
( timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timing ) > gpurun_out/bench_rep$i.log 2>&1; echo "bench$i rc=$?"; grep "bench\] timed" gpurun_out/bench_rep$i.log
done
( timeout 900 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|^E  " gpurun_out/pytest_gpu.log | tail -12
( timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline ) > gpurun_out/bench_graph.log 2>&1; echo "bench graph rc=$?"
tail -c 2600 gpurun_out/bench_graph.log | head -c 2500; echo
