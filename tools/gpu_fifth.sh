#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python tools/ab_check.py ) > gpurun_out/ab_check.log 2>&1; echo "ab rc=$?"
cat gpurun_out/ab_check.log | grep -v Warning | tail -40
