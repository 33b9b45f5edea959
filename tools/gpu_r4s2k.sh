#!/bin/bash
# round 4 (second session): SSIM kernels with branch-free loads: parity, VALU instructions per launch (SQ counters), kernel durations in the replayed step
TAG=${1:-r4s2k}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "ssim or photo or loss or consensus or golden or headline" > gpurun_out/pytest_$TAG.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_$TAG.log
cd /tmp
rm -rf /tmp/pmc_sq
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc_sq -o run -- python $R/bench.py --no-graph --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/pmc_sq_$TAG.log 2>&1; echo "pmc sq rc=$?"
F=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1)
cd $R
PMC_ROWS=200 python tools/pmc_sq.py "$F" > gpurun_out/pmc_sq_$TAG.txt 2>&1; head -1 gpurun_out/pmc_sq_$TAG.txt; grep -E "ssim|edge_smooth|warp" gpurun_out/pmc_sq_$TAG.txt
bash tools/gpu_prof.sh $TAG > gpurun_out/prof_$TAG.out 2>&1; head -2 gpurun_out/step_trace_$TAG.txt; grep -E "ssim" gpurun_out/step_trace_$TAG.txt
