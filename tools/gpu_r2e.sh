#!/bin/bash
# tests + same-box A/B of the 8x16 conv tile + trace + the other benchmark configurations (freeze variant, C5 shape, C2)
TAG=${1:-r02e}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests -m gpu -q -x -s ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|^E  |Error" gpurun_out/pytest_gpu_$TAG.log | tail -12
for V in default CC_CONV_NO_TW16 default; do
  if [ $V = default ]; then E=""; else E="$V=1"; fi
  ( env $E CC_BENCH_DETAIL=gpurun_out/calls_${TAG}_$V.txt timeout 300 python bench.py --no-cpu-baseline --steps 30 ) > gpurun_out/bench_${TAG}_$V.log 2> gpurun_out/bench_${TAG}_$V.err
  echo "$V: $(grep timed gpurun_out/bench_${TAG}_$V.err)"
done
bash tools/gpu_prof.sh $TAG > gpurun_out/prof_$TAG.out 2>&1; head -40 gpurun_out/step_trace_$TAG.txt
( timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --freeze ) > gpurun_out/bench_${TAG}_freeze.log 2> gpurun_out/bench_${TAG}_freeze.err; echo "freeze: $(grep timed gpurun_out/bench_${TAG}_freeze.err)"
( timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --config c2 ) > gpurun_out/bench_${TAG}_c2.log 2> gpurun_out/bench_${TAG}_c2.err; echo "c2: $(grep timed gpurun_out/bench_${TAG}_c2.err)"
( timeout 400 python bench.py --no-cpu-baseline --no-kernel-timing --height 512 --width 1664 --batch 2 ) > gpurun_out/bench_${TAG}_c5.log 2> gpurun_out/bench_${TAG}_c5.err; echo "c5 shape: $(grep timed gpurun_out/bench_${TAG}_c5.err)"
( timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --batch 8 ) > gpurun_out/bench_${TAG}_b8.log 2> gpurun_out/bench_${TAG}_b8.err; echo "b8: $(grep timed gpurun_out/bench_${TAG}_b8.err)"
