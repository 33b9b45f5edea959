#!/bin/bash
# round 4 (second session): kernel trace of tools/head_dgrad_probe.py -- which kernels one small-map data-gradient call launches
mkdir -p gpurun_out
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_hdp -o hdp -- python /root/repo/tools/head_dgrad_probe.py ) > gpurun_out/hdp.log 2>&1
F=$(find gpurun_out/prof_hdp -name "*kernel_stats.csv" | head -1)
head -20 "$F"
