#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( CC_FORCE_COMM=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 5 ) > gpurun_out/bench_r3z2_comm.log 2> gpurun_out/bench_r3z2_comm.err; echo "comm rc=$? $(grep timed gpurun_out/bench_r3z2_comm.err)"
( timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 5 ) > gpurun_out/bench_r3z2_plain.log 2> gpurun_out/bench_r3z2_plain.err; echo "plain rc=$? $(grep timed gpurun_out/bench_r3z2_plain.err)"
python - <<'PY'
import json
for m in ("comm","plain"):
    for l in open('gpurun_out/bench_r3z2_%s.log'%m):
        if l.startswith('{'):
            d=json.loads(l); print(m, d['value'], d['ms_per_step'], d['step_ms']['median'], d['config'].get('hip_runtime'), json.dumps(d.get('comm'))[:300] if d.get('comm') else None)
PY
