#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/wgrad_sweep.py > gpurun_out/wgrad_sweep.log 2>&1; echo "sweep rc=$?"
tail -5 gpurun_out/wgrad_sweep.log
