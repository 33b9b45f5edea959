#!/bin/bash
# tail-quantisation probe + PMC traffic refresh
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 200 python tools/tail_probe.py ) > gpurun_out/tail_probe.txt 2>&1; echo "tail rc=$?"; cat gpurun_out/tail_probe.txt | tail -20
bash tools/gpu_pmc2.sh > gpurun_out/pmc2_g.out 2>&1; tail -6 gpurun_out/pmc2_g.out
