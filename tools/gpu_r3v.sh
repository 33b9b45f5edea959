#!/bin/bash
# A/B of alternate library builds (tools/_bin/libccengine_<tag>.so), same box: edit the list in the for loop
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_nets_gpu.py -m gpu -q -x ) > gpurun_out/pytest_r3v.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/pytest_r3v.log
for L in prev tools prev tools; do
  ( CC_LIB_PATH=$PWD/tools/_bin/libccengine_$L.so timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 5 ) > gpurun_out/bench_r3v_$L.log 2> gpurun_out/bench_r3v_$L.err || true
  echo "$L: $(grep timed gpurun_out/bench_r3v_$L.err)"
done
