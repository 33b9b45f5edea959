#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv" 2>&1 | tail -3
export CC_LIB_PATH=$PWD/tools/_bin/libccengine_tools.so
for V in 1 0; do
CC_CONV_STACK=$V timeout 300 python bench.py --no-cpu-baseline --steps 10 > gpurun_out/r4p_k$V.log 2>&1
done
python - <<'PY'
import json
for V in (1,0):
    for ln in open('gpurun_out/r4p_k%d.log'%V):
        if ln.startswith('{'):
            d=json.loads(ln); r=d['roofline']
            print(V, d['ms_per_step'], r['conv_family']['ms_per_step'], r['mfma_executed']['ms_per_step'])
            for k,v in r['by_kernel'].items():
                if 'conv_patch' in k: print('   %-40s %s'%(k[:40],v))
PY
unset CC_LIB_PATH
bash tools/gpu_ab_env.sh r4p CC_CONV_STACK=0
