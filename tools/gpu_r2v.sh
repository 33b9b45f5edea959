#!/bin/bash
TAG=${1:-r02v}
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_nets_gpu.py tests/test_headline_gpu.py -m gpu -q -x ) > gpurun_out/pytest_k_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|^E  |Error" gpurun_out/pytest_k_$TAG.log | tail -12
for i in 1 2; do
( timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 ) > gpurun_out/bench_${TAG}_$i.log 2> gpurun_out/bench_${TAG}_$i.err
echo "run $i: $(grep timed gpurun_out/bench_${TAG}_$i.err)"
done
bash tools/gpu_prof.sh $TAG > gpurun_out/prof_$TAG.out 2>&1; grep -E "k_conv_patch|k_wgrad<|one replayed" gpurun_out/step_trace_$TAG.txt
