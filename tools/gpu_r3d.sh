#!/bin/bash
# round 3: GPU parity suite + the default bench line (tools-build kernel timing, CPU baseline on the reference archive)
TAG=${1:-r3d}
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|^E  " gpurun_out/pytest_gpu_$TAG.log | tail -12
( timeout 600 python bench.py ) > gpurun_out/bench_$TAG.log 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"
grep -E "cpu baseline|timed|tools build|reference" gpurun_out/bench_$TAG.err | tail -8
python - <<PY
import json
for l in open('gpurun_out/bench_$TAG.log'):
    if l.startswith('{'):
        d=json.loads(l)
        print(d['value'], d['ms_per_step'], d['step_ms'])
        cb=d.get('cpu_baseline',{}); print({k:cb.get(k) for k in ('value','kind','cores','s_per_step')}); print(d.get('parity',{}).get('loss_rel'), d.get('parity',{}).get('ok'))
        r=d['roofline'] or {}; print({k:v for k,v in r.items() if k not in ('by_kernel','by_call_group','timing','conv_family')}); print(r.get('conv_family'))
        print({k:v for k,v in (d['kernels'] or {}).items() if 'gbps' in v})
PY
