#!/bin/bash
# first GPU contact: smoke, GPU parity tests, short bench.  Each stage under its own timeout.
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > gpurun_out/device.txt
nproc >> gpurun_out/device.txt; lscpu | grep -m1 "Model name" >> gpurun_out/device.txt
( time timeout 600 python __graft_entry__.py smoke ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" 
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
( time timeout 600 python bench.py --steps 5 --warmup 2 ) > gpurun_out/bench_first.log 2>&1; echo "bench rc=$?"
tail -5 gpurun_out/smoke.log; tail -15 gpurun_out/pytest_gpu.log; tail -c 3000 gpurun_out/bench_first.log
