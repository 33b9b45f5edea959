#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 500 python tools/nan_hunt.py ) > gpurun_out/nan_hunt.log 2>&1; echo "rc=$?"
grep -v Warning gpurun_out/nan_hunt.log | tail -50
