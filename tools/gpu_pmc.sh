#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|-)?.*(MFMA|SQ_WAIT|SQ_BUSY_CY|SQ_WAVE_CYCLES|SQ_INSTS_LDS|LDS_BANK|SQ_ACTIVE_INST|SQ_INSTS_VALU |SQ_WAVES|GRBM_GUI_ACTIVE|SQ_INST_CYCLES|VALUBusy|MfmaUtil|SQ_LDS_IDX)" | head -60 > /root/repo/gpurun_out/counters.txt
wc -l /root/repo/gpurun_out/counters.txt; head -40 /root/repo/gpurun_out/counters.txt
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; do
  CC_NO_WGRAD3X3=0 timeout 120 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc -o run -- python /root/repo/tools/wgrad_ablate.py b2f128 > /root/repo/gpurun_out/pmc_run.log 2>&1
  F=$(find /root/repo/gpurun_out/pmc -name "*counter_collection.csv" | head -1)
  python - "$F" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    if "wgrad3x3" in k or "conv_patch" in k:
        print(k, {c: "%.3g" % (v / max(cnt[(k, c)], 1)) for c, v in agg[k].items()})
PY
  rm -rf /root/repo/gpurun_out/pmc
done
tail -2 /root/repo/gpurun_out/pmc_run.log
