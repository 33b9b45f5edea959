#!/bin/bash
# round 3: GPU parity suite + bench (no CPU leg) + rocprofv3 kernel stats / one-step trace of the current build
TAG=${1:-r3b}
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|^E  " gpurun_out/pytest_gpu_$TAG.log | tail -12
( CC_BENCH_DETAIL=gpurun_out/calls_$TAG.txt timeout 300 python bench.py --no-cpu-baseline ) > gpurun_out/bench_${TAG}_nocpu.log 2> gpurun_out/bench_${TAG}_nocpu.err; echo "bench(nocpu) rc=$?"
tail -3 gpurun_out/bench_${TAG}_nocpu.err
bash tools/gpu_prof.sh $TAG > gpurun_out/prof_$TAG.out 2>&1; head -70 gpurun_out/step_trace_$TAG.txt
