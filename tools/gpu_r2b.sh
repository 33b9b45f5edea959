#!/bin/bash
# GPU parity tests, bench line without and with the (bounded) CPU baseline, per-call detail, rocprofv3 kernel stats + step trace
TAG=${1:-r02b}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests -m gpu -q -x -s ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|^E  |Error" gpurun_out/pytest_gpu_$TAG.log | tail -12
( CC_BENCH_DETAIL=gpurun_out/calls_$TAG.txt timeout 300 python bench.py --no-cpu-baseline ) > gpurun_out/bench_${TAG}_nocpu.log 2> gpurun_out/bench_${TAG}_nocpu.err; echo "bench(nocpu) rc=$?"
tail -2 gpurun_out/bench_${TAG}_nocpu.err
bash tools/gpu_prof.sh $TAG > gpurun_out/prof_$TAG.out 2>&1; head -45 gpurun_out/step_trace_$TAG.txt
( timeout 420 python bench.py --no-kernel-timing ) > gpurun_out/bench_$TAG.log 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"
grep -E "cpu baseline|timed" gpurun_out/bench_$TAG.err | tail -6
python - <<'PY'
import json,sys
for f in sys.argv[1:]:
    pass
for name in ("gpurun_out/bench_%s.log" % "TAGX",):
    pass
PY
python -c "
import json
for l in open('gpurun_out/bench_$TAG.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d.get('cpu_baseline'), d.get('parity'))
for l in open('gpurun_out/bench_${TAG}_nocpu.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print({k:v for k,v in r.items() if k!='by_kernel'}); print(r['by_kernel'])
"
