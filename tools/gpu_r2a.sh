#!/bin/bash
# round-2 first GPU pass: parity tests (incl. headline sizes), smoke, default bench line (CPU baseline + parity gate),
# rocprofv3 kernel stats + one-step trace of the replayed graph
TAG=${1:-r02a}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests -m gpu -q -x -s ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|^E  |headline parity|worst small" gpurun_out/pytest_gpu_$TAG.log | tail -30
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke_$TAG.log
( timeout 900 python bench.py ) > gpurun_out/bench_$TAG.log 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"
tail -4 gpurun_out/bench_$TAG.err
bash tools/gpu_prof.sh $TAG > gpurun_out/prof_$TAG.out 2>&1; head -40 gpurun_out/step_trace_$TAG.txt
