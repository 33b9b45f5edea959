#!/bin/bash
# rocprofv3 kernel stats + one-step trace + per-call eager detail of the current build
TAG=${1:-r3h}
mkdir -p gpurun_out
export TMPDIR=/tmp
( CC_BENCH_DETAIL=gpurun_out/calls_$TAG.txt timeout 300 python bench.py --no-cpu-baseline ) > gpurun_out/bench_${TAG}_nocpu.log 2> gpurun_out/bench_${TAG}_nocpu.err; echo "bench(nocpu) rc=$?"
tail -2 gpurun_out/bench_${TAG}_nocpu.err
bash tools/gpu_prof.sh $TAG > gpurun_out/prof_$TAG.out 2>&1; head -5 gpurun_out/step_trace_$TAG.txt
