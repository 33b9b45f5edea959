#!/bin/bash
# tests + same-box A/B of the BM=16 conv tile and the deferred grouped weight gradients + trace + SQ counters of the conv kernels
TAG=${1:-r02d}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests -m gpu -q -x -s ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|^E  |Error" gpurun_out/pytest_gpu_$TAG.log | tail -12
for V in default CC_CONV_NO_BM16 CC_NO_WGRAD_QUEUE default; do
  if [ $V = default ]; then E=""; else E="$V=1"; fi
  ( env $E CC_BENCH_DETAIL=gpurun_out/calls_${TAG}_$V.txt timeout 300 python bench.py --no-cpu-baseline --steps 30 ) > gpurun_out/bench_${TAG}_$V.log 2> gpurun_out/bench_${TAG}_$V.err
  echo "$V: $(grep timed gpurun_out/bench_${TAG}_$V.err)"
done
bash tools/gpu_prof.sh $TAG > gpurun_out/prof_$TAG.out 2>&1; head -50 gpurun_out/step_trace_$TAG.txt
cd /tmp
rm -rf /tmp/pmc_sq
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d /tmp/pmc_sq -o run -- python $R/bench.py --no-graph --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/pmc_sq_$TAG.log 2>&1; echo "pmc sq rc=$?"
F=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1)
cd $R
python tools/pmc_sq.py "$F" > gpurun_out/pmc_sq_$TAG.txt 2>&1; cat gpurun_out/pmc_sq_$TAG.txt
