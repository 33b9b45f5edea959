#!/bin/bash
# A/B of alternate library builds (tools/_bin/libccengine_<tag>.so), same box: edit the list in the for loop
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for L in slack0 tools slack0 tools; do
  ( CC_LIB_PATH=$PWD/tools/_bin/libccengine_$L.so timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 5 ) > gpurun_out/bench_r3v_$L.log 2> gpurun_out/bench_r3v_$L.err || true
  echo "$L: $(grep timed gpurun_out/bench_r3v_$L.err)"
done
( CC_LIB_PATH=$PWD/tools/_bin/libccengine_tools.so timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv" ) > gpurun_out/pytest_r3v.log 2>&1; echo "pytest(conv) rc=$?"; tail -2 gpurun_out/pytest_r3v.log
