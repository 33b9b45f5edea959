#!/bin/bash
# round 3: conv / net parity subset on the new epilogue, then same-box A/B: previous tools build vs the current one
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -k "conv or net_forward or full_step or headline" ) > gpurun_out/pytest_gpu_r3j.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|^E  " gpurun_out/pytest_gpu_r3j.log | tail -8
ABL_LIST="tools prev tools prev" bash tools/ablate_conv.sh run
