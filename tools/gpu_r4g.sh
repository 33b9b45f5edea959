#!/bin/bash
# round 4: per-layer-shape table of one eager step (tools build, CC_TIMING_DETAIL) + rocprofv3 one-step trace of the graph replay
TAG=${1:-r4g}
mkdir -p gpurun_out
export TMPDIR=/tmp
CC_TIMING_DETAIL=1 CC_TIMING_DUMP=gpurun_out/layers_$TAG.tsv timeout 600 python bench.py --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$TAG.log 2> gpurun_out/bench_$TAG.err
python tools/layer_rates.py gpurun_out/layers_$TAG.tsv > gpurun_out/layer_rates_$TAG.txt
head -70 gpurun_out/layer_rates_$TAG.txt
bash tools/gpu_prof.sh $TAG > gpurun_out/prof_$TAG.out 2>&1
head -75 gpurun_out/step_trace_$TAG.txt
