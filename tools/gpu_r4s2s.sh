#!/bin/bash
# round 4 (second session): bench.py's gradient-vector parity figure (gradient_l2_rel against the CPU reference) with and without this session's kernels
mkdir -p gpurun_out
export TMPDIR=/tmp
export CC_LIB_PATH=$PWD/tools/_bin/libccengine_tools.so
( timeout 400 python bench.py --steps 5 --warmup 2 --no-kernel-timing ) > gpurun_out/bench_r4s2s_a.log 2> gpurun_out/bench_r4s2s_a.err
( CC_NO_HEAD_KERNELS=1 CC_NO_HEAD_ACC=1 CC_NO_WINO_WGRAD_LIST=1 CC_WGRAD_PARK_TARGET=512 CC_WGRAD_PARK_MINRANGE=32 timeout 400 python bench.py --steps 5 --warmup 2 --no-kernel-timing ) > gpurun_out/bench_r4s2s_b.log 2> gpurun_out/bench_r4s2s_b.err
( CC_NO_WINO=1 CC_NO_WINO_WGRAD=1 timeout 400 python bench.py --steps 5 --warmup 2 --no-kernel-timing ) > gpurun_out/bench_r4s2s_c.log 2> gpurun_out/bench_r4s2s_c.err
python - <<PY
import json
for t in 'abc':
    for l in open('gpurun_out/bench_r4s2s_%s.log' % t):
        if l.startswith('{'):
            d=json.loads(l); p=d['parity']; print(t, d['ms_per_step'], p.get('gradient_l2_rel'), p.get('grad_norm_rel'), p.get('loss_rel'), p.get('update'))
PY
