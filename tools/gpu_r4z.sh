#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "thin or wgrad or conv" 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4z_bench.log 2>&1
grep -E "timed" gpurun_out/r4z_bench.log
python - <<'PY'
import json
for ln in open('gpurun_out/r4z_bench.log'):
    if ln.startswith('{'):
        d=json.loads(ln); r=d['roofline']
        for k,v in r['by_kernel'].items(): print('   %-40s %s'%(k[:40],v))
PY
