#!/bin/bash
# rocprofv3 kernel trace of the bench command -> per-kernel CSV summary in gpurun_out/prof_<tag>/
TAG=${1:-r01}
mkdir -p gpurun_out
export TMPDIR=/tmp
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_$TAG -o $TAG -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing ) > gpurun_out/rocprof_$TAG.log 2>&1; echo "rocprof rc=$?"
find gpurun_out/prof_$TAG -name "*kernel_stats*" | head -3
F=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
T=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
python tools/step_trace.py "$T" > gpurun_out/step_trace_$TAG.txt 2>&1; head -60 gpurun_out/step_trace_$TAG.txt
# the bulky per-dispatch trace does not need to travel back
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete
grep "bench\]" gpurun_out/rocprof_$TAG.log
