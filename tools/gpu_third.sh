#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 400 python tools/conv_bench.py ) > gpurun_out/conv_bench.log 2>&1; echo "conv_bench rc=$?"
( timeout 900 python -m pytest tests -m gpu -q -s ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
( timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline ) > gpurun_out/bench_graph.log 2>&1; echo "bench graph rc=$?"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof -o r01 -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing ) > gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?"
( timeout 240 python - <<'PY'
import sys, time, os, torch
sys.path.insert(0, '/root/repo')
import bench
from cc_amd import synthetic as syn
class A: pass
a = A(); a.config='c3'; a.cpu_steps=1; a.cpu_threads=64
t=time.time(); print(bench.cpu_baseline(syn.sample(4,256,832,seed=1,smooth=3), a), time.time()-t)
PY
) > gpurun_out/cpu_baseline.log 2>&1; echo "cpu baseline rc=$?"
grep -E "passed|failed|FAILED" gpurun_out/pytest_gpu.log | tail -8
grep -E "bench\]" gpurun_out/bench_graph.log; tail -c 1800 gpurun_out/bench_graph.log | head -c 1700; echo
cat gpurun_out/conv_bench.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-28s err %.1e fwd %6.3f ms %5.1f TF (miopen %5.1f) | f+b %6.3f ms %5.1f TF (miopen %5.1f)' % (d['layer'], d['relerr'], d['hip_fwd_ms'], d['hip_fwd_tf'], d['miopen_fwd_tf'], d['hip_fwdbwd_ms'], d['hip_fwdbwd_tf'], d['miopen_fwdbwd_tf']))
    else: print(l.strip())
"
tail -3 gpurun_out/cpu_baseline.log
ls gpurun_out/prof 2>/dev/null | head; find gpurun_out/prof -name "*stats*" | head
