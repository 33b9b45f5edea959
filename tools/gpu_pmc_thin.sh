#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  rm -rf /tmp/pmc3
  PYTHONPATH=$R timeout 120 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/pmc3 -o run -- python $R/tools/thin_probe.py 10 > $R/gpurun_out/pmc3_run.log 2>&1
  F=$(find /tmp/pmc3 -name "*counter_collection.csv" | head -1)
  python - "$F" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    if "wgrad_thin" in k:
        print(k[:60], {c: "%.4g" % (v / max(cnt[(k, c)], 1)) for c, v in agg[k].items()})
PY
done
tail -3 $R/gpurun_out/pmc3_run.log
