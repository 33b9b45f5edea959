#!/bin/bash
# round 4 artefacts on one box: full GPU suite, smoke(), the forced-communication bench line (1-rank RCCL group), rocprofv3 kernel
# stats + one-step trace, PMC traffic passes (-> profiles/pmc_traffic.json), an SQ counter pass, the default bench line
TAG=${1:-r4fin}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|^E  " gpurun_out/pytest_gpu_$TAG.log | tail -8
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke_$TAG.log
( CC_FORCE_COMM=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 ) > gpurun_out/bench_${TAG}_comm.log 2> gpurun_out/bench_${TAG}_comm.err; echo "bench(comm) rc=$?"
python - <<PY
import json
for l in open('gpurun_out/bench_${TAG}_comm.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['step_ms']['median']); print(json.dumps(d['comm'])[:900]); print(d['config'].get('rank_losses'), d['config'].get('rccl_ranks'))
PY
bash tools/gpu_prof.sh $TAG > gpurun_out/prof_$TAG.out 2>&1; head -4 gpurun_out/step_trace_$TAG.txt
bash tools/gpu_pmc2.sh > gpurun_out/pmc2_$TAG.out 2>&1; tail -4 gpurun_out/pmc2_$TAG.out
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
rm -rf /tmp/pmc_sq; cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc_sq -o run -- python $R/bench.py --no-graph --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/pmc_sq_$TAG.log 2>&1; echo "pmc sq rc=$?"
cd $R
F=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1)
PMC_ROWS=60 python tools/pmc_sq.py "$F" > gpurun_out/pmc_sq_$TAG.txt 2>&1; grep -E "^kernel|ssim|warp_" gpurun_out/pmc_sq_$TAG.txt | cut -c1-200
( timeout 600 python bench.py ) > gpurun_out/bench_$TAG.log 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"
grep -E "cpu baseline|timed" gpurun_out/bench_$TAG.err | tail -4
python - <<PY
import json
for l in open('gpurun_out/bench_$TAG.log'):
    if l.startswith('{'):
        d=json.loads(l)
        print(d['value'], d['ms_per_step'], d['step_ms'])
        cb=d.get('cpu_baseline',{}); print({k:cb.get(k) for k in ('value','kind','cores','s_per_step')}); print(d.get('parity',{}).get('loss_rel'), d.get('parity',{}).get('ok'))
        r=d['roofline'] or {}; print({k:v for k,v in r.items() if k not in ('by_kernel','by_call_group','timing','conv_family')}); print(r.get('conv_family'))
        print({k:(v.get('frac'),v.get('avg_us')) for k,v in (d['kernels'] or {}).items() if 'gbps' in v})
PY
