#!/bin/bash
export TMPDIR=/tmp
cd /tmp
for V in 0; do
CC_NO_WGRAD3X3=$V timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$V -o run -- python /root/repo/tools/wgrad_ablate.py b2f128 > /tmp/kt$V.log 2>&1
F=$(find /tmp/kt$V -name "*kernel_stats.csv" | head -1)
echo "== CC_NO_WGRAD3X3=$V"; grep -E "anonymous|Name" "$F" | cut -d, -f1-4 | cut -c1-150 | head -8
done
