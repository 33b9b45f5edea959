#!/bin/bash
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
for CFG in "CC_WGRAD_THIN_DBG=0" "CC_WGRAD_THIN_DBG=1" "CC_WGRAD_THIN_DBG=0 CC_WGRAD_THIN_NOSWZ=1" "CC_WGRAD_THIN_DBG=1 CC_WGRAD_THIN_NOSWZ=1"; do
  rm -rf /tmp/ab
  env $CFG PYTHONPATH=$R timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab -o run -- python $R/tools/thin_probe.py 20 > /dev/null 2>&1
  F=$(find /tmp/ab -name "*kernel_stats.csv" | head -1)
  python - "$F" "$CFG" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "wgrad_thin<" in r["Name"]:
        print("%-50s calls %s avg %.1f us" % (sys.argv[2], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
