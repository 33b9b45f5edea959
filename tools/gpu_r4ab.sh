#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
export CC_LIB_PATH=$PWD/tools/_bin/libccengine_tools.so
for V in 0 1 2; do
CC_WGRAD_THIN_DBG=$V timeout 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/r4ab_$V.log 2>&1
python - <<PY
import json
for ln in open('gpurun_out/r4ab_$V.log'):
    if ln.startswith('{'):
        d=json.loads(ln); r=d['roofline']
        print($V, d['ms_per_step'], {k:v['ms'] for k,v in r['by_kernel'].items() if 'thin' in k})
PY
done
