#!/bin/bash
# refresh profiles/pmc_traffic.json for the current library (two PMC passes) and print the default bench line
TAG=${1:-pmcb}
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu_pmc2.sh > gpurun_out/pmc2_$TAG.out 2>&1; tail -3 gpurun_out/pmc2_$TAG.out
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
( timeout 420 python bench.py ) > gpurun_out/bench_$TAG.log 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"
grep -E "timed|cpu baseline" gpurun_out/bench_$TAG.err | tail -5
python -c "
import json
for l in open('gpurun_out/bench_$TAG.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d.get('cpu_baseline',{}).get('value'), d.get('parity',{}).get('loss_rel'))
        r=d['roofline']; print({k:v for k,v in r.items() if k in ('kernel','achieved','frac','traffic','launches','avg_launch_us')})
"
