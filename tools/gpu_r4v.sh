#!/bin/bash
# SSIM kernels after the packed-operand rewrite: parity, VALU instructions per pixel (SQ counters), step time
TAG=${1:-r4v}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "ssim or photo or loss or consensus or golden or headline" 2>&1 | tail -3
cd /tmp
rm -rf /tmp/pmc_sq
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc_sq -o run -- python $R/bench.py --no-graph --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/pmc_sq_$TAG.log 2>&1; echo "pmc sq rc=$?"
F=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1)
cd $R
PMC_ROWS=200 python tools/pmc_sq.py "$F" > gpurun_out/pmc_sq_$TAG.txt 2>&1; head -1 gpurun_out/pmc_sq_$TAG.txt; grep -E "ssim|edge_smooth|warp" gpurun_out/pmc_sq_$TAG.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench.log 2>&1
grep -E "timed" gpurun_out/${TAG}_bench.log
python - <<PY
import json
for ln in open('gpurun_out/${TAG}_bench.log'):
    if ln.startswith('{'):
        d=json.loads(ln)
        for k,v in d.get('kernels',{}).items():
            if 'ssim' in k or 'warp' in k or 'smooth' in k: print(k, v)
PY
