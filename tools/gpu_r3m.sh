#!/bin/bash
# single-graph step vs the forced-communication (two-graph) step on one box
mkdir -p gpurun_out
export TMPDIR=/tmp
for V in plain comm plain comm; do
  if [ $V = comm ]; then E="CC_FORCE_COMM=1"; else E="X=1"; fi
  ( env $E timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 5 ) > gpurun_out/bench_r3m_$V.log 2> gpurun_out/bench_r3m_$V.err
  python - <<PY
import json
for l in open('gpurun_out/bench_r3m_$V.log'):
    if l.startswith('{'):
        d=json.loads(l); print('$V', d['ms_per_step'], d['step_ms']['median'], d['step_ms']['p10'], d['step_ms']['p90'], (d.get('comm') or {}).get('exposed_ms'))
PY
done
