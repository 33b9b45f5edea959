#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline ) > gpurun_out/bench_graph.log 2>&1; echo "bench rc=$?"
grep "bench\]" gpurun_out/bench_graph.log
python - <<'PY'
import json
for l in open('gpurun_out/bench_graph.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline'])
        for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms']): print('  %-22s calls %4d ms %8.3f %s' % (k, v['calls'], v['ms'], v.get('tflops', v.get('gbps'))))
PY
