#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 400 python tools/conv_bench.py ) > gpurun_out/conv_bench.log 2>&1; echo "conv_bench rc=$?"
( timeout 900 python -m pytest tests -m gpu -q -s ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
( timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline ) > gpurun_out/bench_graph.log 2>&1; echo "bench graph rc=$?"
grep -E "passed|failed|FAILED|^E  " gpurun_out/pytest_gpu.log | tail -12
grep -E "bench\]" gpurun_out/bench_graph.log; tail -c 2500 gpurun_out/bench_graph.log | head -c 2400; echo
cat gpurun_out/conv_bench.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-28s err %.1e fwd %6.3f ms %5.1f TF (miopen %5.1f) | f+b %6.3f ms %5.1f TF (miopen %5.1f)' % (d['layer'], d['relerr'], d['hip_fwd_ms'], d['hip_fwd_tf'], d['miopen_fwd_tf'], d['hip_fwdbwd_ms'], d['hip_fwdbwd_tf'], d['miopen_fwdbwd_tf']))
    else: print(l.strip())
"
