#!/bin/bash
# bias-gradient table: parity tests, same-box A/B against one pass per layer, one-step trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_nets_gpu.py -m gpu -q -x ) > gpurun_out/pytest_r3s.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_r3s.log
for V in default CC_NO_BIAS_TABLE=1 default CC_NO_BIAS_TABLE=1; do
  if [ "$V" = default ]; then E=""; else E="$V"; fi
  ( env $E timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 ) > gpurun_out/bench_r3s_$V.log 2> gpurun_out/bench_r3s_$V.err
  echo "$V: $(grep timed gpurun_out/bench_r3s_$V.err)"
done
bash tools/gpu_prof.sh r3s > gpurun_out/prof_r3s.out 2>&1; head -3 gpurun_out/step_trace_r3s.txt; grep -E "k_act_bwd|k_bias_table|k_wgrad_reduce_table|elementwise|k_scale" gpurun_out/step_trace_r3s.txt
