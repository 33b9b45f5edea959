#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
CC_TIMING_DETAIL=1 CC_TIMING_DUMP=gpurun_out/layers_r4ag.tsv timeout 600 python bench.py --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r4ag.log 2> gpurun_out/bench_r4ag.err
python tools/layer_rates.py gpurun_out/layers_r4ag.tsv > gpurun_out/layer_rates_r4ag.txt
grep -E "pad|k_wgrad<" gpurun_out/layer_rates_r4ag.txt | head -40 | cut -c1-140
