#!/bin/bash
# L1 (TCP) / L2 (TCC) hit counters of one eager step, per kernel (the gather-heavy warp / cost-volume kernels are the point)
TAG=${1:-cache}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for SET in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $SET | cut -d' ' -f1)
  rm -rf /tmp/pmc_$N
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/pmc_$N -o run -- python $R/bench.py --no-graph --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/pmc_${N}_$TAG.log 2>&1; echo "pmc $N rc=$?"
  F=$(find /tmp/pmc_$N -name "*counter_collection.csv" | head -1)
  ( cd $R && PMC_ROWS=400 python tools/pmc_sq.py "$F" > gpurun_out/pmc_${N}_$TAG.txt 2>&1 )
  grep -E "warp|corr_fwd4|ssim_photo" $R/gpurun_out/pmc_${N}_$TAG.txt | head -8
done
