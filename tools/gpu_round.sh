#!/bin/bash
# PMC traffic passes + the full default bench (with kernel timing, no cpu baseline)
bash tools/gpu_pmc2.sh
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
( timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline ) > gpurun_out/bench_graph.log 2>&1; echo "bench rc=$?"
grep -E "bench\]" gpurun_out/bench_graph.log | tail -1
python - <<'PY'
import json
for l in open('gpurun_out/bench_graph.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['value'], d['ms_per_step']); print({k:v for k,v in r.items() if k!='by_kernel'})
        for k,v in r['by_kernel'].items(): print('  ',k,v)
PY
