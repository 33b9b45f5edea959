#!/bin/bash
# N-ary gradient sums: parity tests + same-box A/B (CC_NO_SUM_N=1: pairwise adds)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_nets_gpu.py -m gpu -q -x ) > gpurun_out/pytest_r3w.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_r3w.log
for V in default CC_NO_SUM_N=1 default CC_NO_SUM_N=1; do
  if [ "$V" = default ]; then E=""; else E="$V"; fi
  ( env $E timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 ) > gpurun_out/bench_r3w_$V.log 2> gpurun_out/bench_r3w_$V.err
  echo "$V: $(grep timed gpurun_out/bench_r3w_$V.err)"
done
