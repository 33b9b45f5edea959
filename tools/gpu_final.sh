#!/bin/bash
# Everything the round's artefacts come from, on one box: GPU parity tests, smoke(), rocprofv3 kernel stats + one-step trace,
# the two PMC traffic passes, and the default bench line (with the CPU baseline).  TAG names the files under gpurun_out/.
TAG=${1:-r01f}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
bash tools/gpu_prof.sh $TAG > gpurun_out/prof_$TAG.out 2>&1; head -12 gpurun_out/step_trace_$TAG.txt
bash tools/gpu_pmc2.sh > gpurun_out/pmc2.out 2>&1; tail -6 gpurun_out/pmc2.out
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
( timeout 600 python bench.py ) > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; echo "bench rc=$?"
tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
for l in open('gpurun_out/bench_default.log'):
    if l.startswith('{'):
        d = json.loads(l)
        print(d['value'], d['unit'], d['ms_per_step'], 'ms/step')
        print({k: v for k, v in d['roofline'].items() if k != 'by_kernel'})
        print(d.get('cpu_baseline'))
PY
