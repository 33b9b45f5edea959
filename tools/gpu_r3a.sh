#!/bin/bash
# round 3, first call: GPU parity suite on the generalised class launcher + the cross-network merge probe
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -x ) > gpurun_out/pytest_gpu_r3a.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|^E  " gpurun_out/pytest_gpu_r3a.log | tail -12
( timeout 600 python tools/merge_probe.py ) > gpurun_out/merge_probe_r3a.log 2>&1; echo "probe rc=$?"
tail -25 gpurun_out/merge_probe_r3a.log
( timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing ) > gpurun_out/bench_r3a.log 2>&1; echo "bench rc=$?"
grep -E "bench\]|^\{" gpurun_out/bench_r3a.log | tail -3 | cut -c1-400
