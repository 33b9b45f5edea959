#!/bin/bash
# tests + bench (no CPU leg) + rocprofv3 stats/trace + PMC traffic passes + default bench (with the bounded CPU baseline)
TAG=${1:-r02c}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests -m gpu -q -x -s ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|^E  |Error" gpurun_out/pytest_gpu_$TAG.log | tail -12
( CC_BENCH_DETAIL=gpurun_out/calls_$TAG.txt timeout 300 python bench.py --no-cpu-baseline ) > gpurun_out/bench_${TAG}_nocpu.log 2> gpurun_out/bench_${TAG}_nocpu.err; echo "bench(nocpu) rc=$?"
tail -2 gpurun_out/bench_${TAG}_nocpu.err
bash tools/gpu_prof.sh $TAG > gpurun_out/prof_$TAG.out 2>&1; head -60 gpurun_out/step_trace_$TAG.txt
bash tools/gpu_pmc2.sh > gpurun_out/pmc2_$TAG.out 2>&1; tail -8 gpurun_out/pmc2_$TAG.out
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
( timeout 420 python bench.py ) > gpurun_out/bench_$TAG.log 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"
grep -E "cpu baseline|timed" gpurun_out/bench_$TAG.err | tail -6
python -c "
import json
for l in open('gpurun_out/bench_$TAG.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d.get('cpu_baseline',{}).get('value'), d.get('parity',{}).get('loss_rel'))
        r=d['roofline']; print({k:v for k,v in r.items() if k!='by_kernel'}); print(r['by_kernel'])
        print({k:v for k,v in d['kernels'].items() if 'gbps' in v})
"
