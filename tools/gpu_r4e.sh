#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r4e}
rm -f gpurun_out/${TAG}_probe.txt
for A in 0 64 128; do
  CC_WINO_ABL=$A timeout 200 python tools/wino_probe.py 2>/dev/null | tail -1 >> gpurun_out/${TAG}_probe.txt
done
cat gpurun_out/${TAG}_probe.txt
