#!/bin/bash
# A/B: which weight-gradient kernel takes the 64-channel layers at 64x208 (thin / generic / 3x3), then the per-shape table
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/gpu_ab_env.sh r3o "CC_WGRAD_THIN_MAXCOMBO=15" "CC_WGRAD_THIN_MAXCOMBO=15 CC_W3_MINM=64" "CC_W3_MINM=64"
CC_TIMING_DETAIL=1 CC_TIMING_DUMP=gpurun_out/layers_r3o.tsv timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 5 > gpurun_out/bench_r3o.log 2> gpurun_out/bench_r3o.err
python tools/layer_rates.py gpurun_out/layers_r3o.tsv > gpurun_out/layer_rates_r3o.txt
head -30 gpurun_out/layer_rates_r3o.txt
