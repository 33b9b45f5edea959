#!/bin/bash
# full GPU suite + smoke() + SQ-counter pass of the final build
TAG=${1:-r02r}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests -m gpu -q -x ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|^E  |Error" gpurun_out/pytest_gpu_$TAG.log | tail -12
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/smoke_$TAG.log 2>&1; tail -2 gpurun_out/smoke_$TAG.log
cd /tmp
rm -rf /tmp/pmc_sq
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d /tmp/pmc_sq -o run -- python $R/bench.py --no-graph --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/pmc_sq_$TAG.log 2>&1; echo "pmc sq rc=$?"
F=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1)
cd $R
python tools/pmc_sq.py "$F" > gpurun_out/pmc_sq_$TAG.txt 2>&1; head -20 gpurun_out/pmc_sq_$TAG.txt
