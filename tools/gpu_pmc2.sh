#!/bin/bash
# HBM traffic of every kernel of one (eager) training step: two separate PMC passes (FETCH_SIZE, WRITE_SIZE), no trace domains
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --no-graph --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing"
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -o run -- $CMD > $R/gpurun_out/pmc_$C.log 2>&1; echo "pmc $C rc=$?"
done
F=$(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
cd $R
python tools/pmc_traffic.py "$F" "$W" gpurun_out/pmc_traffic.json "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python bench.py --no-graph --steps 1 --warmup 1" | tee gpurun_out/pmc_traffic.txt
