#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
CC_TIMING_DETAIL=1 CC_TIMING_DUMP=gpurun_out/layers_c5.tsv timeout 600 python bench.py --height 512 --width 1664 --batch 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_c5.log 2> gpurun_out/bench_c5.err
python tools/layer_rates.py gpurun_out/layers_c5.tsv > gpurun_out/layer_rates_c5.txt
head -45 gpurun_out/layer_rates_c5.txt | cut -c1-170
tail -25 gpurun_out/layer_rates_c5.txt | cut -c1-170
