#!/bin/bash
# verdict r3 item 7: the N > 1 code path rehearsed on one GPU (a 1-rank RCCL group): --freeze (single segment) and the 1664x512 b=2 shape
mkdir -p gpurun_out
export TMPDIR=/tmp
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
( echo "== CC_FORCE_COMM=1 bench.py --freeze"; CC_FORCE_COMM=1 timeout 600 python bench.py --freeze --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -E "^\{|timed|comm|graph|segment|Error|error" | cut -c1-1500
  echo "== CC_FORCE_COMM=1 bench.py (all networks)"; CC_FORCE_COMM=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -E "^\{|timed|comm|graph|segment|Error|error" | cut -c1-1500
  echo "== CC_FORCE_COMM=1 bench.py --height 512 --width 1664 --batch 2"; CC_FORCE_COMM=1 timeout 900 python bench.py --height 512 --width 1664 --batch 2 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -E "^\{|timed|comm|graph|segment|Error|error" | cut -c1-1500
  echo "== bench.py --height 512 --width 1664 --batch 2 (no comm)"; timeout 900 python bench.py --height 512 --width 1664 --batch 2 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -E "^\{|timed|Error|error" | cut -c1-600
) > gpurun_out/r4t_forced_comm.log 2>&1
cat gpurun_out/r4t_forced_comm.log | cut -c1-400
