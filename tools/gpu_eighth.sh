#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { for i in 1 2 3; do ( timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timing ) > gpurun_out/b.log 2>&1; echo "$1 run$i rc=$? $(grep 'bench\] timed' gpurun_out/b.log)"; done; }
run "memsetfix"
CC_DBG_OLD_WGRAD=1 run "oldwgrad"
CC_DBG_NO_PARITY_SPLIT=1 run "noparitysplit"
