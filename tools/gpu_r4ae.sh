#!/bin/bash
TAG=${1:-r4ae}
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|^E  " gpurun_out/pytest_gpu_$TAG.log | tail -8
( timeout 900 python bench.py ) > gpurun_out/bench_$TAG.log 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"
grep -E "cpu baseline|timed" gpurun_out/bench_$TAG.err | tail -4
python - <<PY
import json
for l in open('gpurun_out/bench_$TAG.log'):
    if l.startswith('{'):
        d=json.loads(l)
        print(d['value'], d['ms_per_step'], d['step_ms'])
        cb=d.get('cpu_baseline',{}); print({k:cb.get(k) for k in ('value','kind','cores','s_per_step')}); print(json.dumps(d.get('parity',{}))[:1200])
PY
