#!/bin/bash
# per-layer-shape kernel rates of one eager step (tools build, CC_TIMING_DETAIL)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CC_TIMING_DETAIL=1 CC_TIMING_DUMP=gpurun_out/layers_r3n.tsv timeout 600 python bench.py --steps 10 --warmup 5 > gpurun_out/bench_r3n.log 2> gpurun_out/bench_r3n.err
tail -1 gpurun_out/bench_r3n.log | cut -c1-300
python tools/layer_rates.py gpurun_out/layers_r3n.tsv > gpurun_out/layer_rates_r3n.txt
head -60 gpurun_out/layer_rates_r3n.txt
