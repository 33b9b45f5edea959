#!/bin/bash
# A/B: loader-wave build of the conv kernels against the tools build (same box), then its per-shape table
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for L in tools lw tools lw; do
  ( CC_LIB_PATH=$PWD/tools/_bin/libccengine_$L.so timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 5 ) > gpurun_out/bench_r3p_$L.log 2> gpurun_out/bench_r3p_$L.err || true
  echo "$L: $(grep timed gpurun_out/bench_r3p_$L.err)"
done
cp tools/_bin/libccengine_tools.so /tmp/tools_keep.so
cp tools/_bin/libccengine_lw.so tools/_bin/libccengine_tools.so      # bench.py's per-kernel pass loads the tools build by name
CC_LIB_PATH=$PWD/tools/_bin/libccengine_lw.so CC_TIMING_DETAIL=1 CC_TIMING_DUMP=gpurun_out/layers_r3p.tsv timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 5 > gpurun_out/bench_r3p.log 2> gpurun_out/bench_r3p.err
cp /tmp/tools_keep.so tools/_bin/libccengine_tools.so
python tools/layer_rates.py gpurun_out/layers_r3p.tsv > gpurun_out/layer_rates_r3p.txt
head -24 gpurun_out/layer_rates_r3p.txt
