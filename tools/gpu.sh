#!/bin/bash
# ONE parametrised GPU-box script (replaces the ~80 one-shot tools/gpu_r*.sh of rounds 2-4):
#     gpurun --timeout 900 -- 'bash tools/gpu.sh TAG step [step ...]'
# steps (run in the order given; every artefact goes to gpurun_out/<kind>_<TAG>...):
#   tests[:EXPR]      pytest -m gpu (optionally -k EXPR)
#   smoke             __graft_entry__.smoke()
#   bench             default bench line (bounded CPU baseline + parity gate + roofline)
#   quick             bench without the CPU leg and without per-kernel timing, 30 steps
#   comm              the forced-communication line (1-rank RCCL group)
#   cfg               other configurations: --freeze, c2, 1664x512 b=2, batch 8
#   prof / prof0      rocprofv3 --kernel-trace --stats of the bench + step_trace table of ONE replayed step (prof0: CC_NET_STREAMS=0)
#   pmc               FETCH_SIZE / WRITE_SIZE passes -> gpurun_out/pmc_traffic.json (+ copy to profiles/)
#   sq                SQ counter pass (tools/pmc_sq.py)
#   mfma              MFMA utilisation of the conv kernels (tools/pmc_mfma.py; run prof0 first for the durations)
#   layers            per-layer-shape table (tools/layer_rates.py)
#   ab:VAR=VAL[,VAR=VAL...]   same-box A/B against the default (tools build of the library; default run first and last)
#   py:SCRIPT[:ARGS]  python tools/SCRIPT ARGS  (probes: wino_bench.py, conv_bench.py, ...)
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
TOOLS_LIB=$R/tools/_bin/libccengine_tools.so
NOCPU="--no-cpu-baseline --no-kernel-timing"

line() {  # print the essentials of a bench JSON line
python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d = json.loads(l)
        print("value %s images/s  %s ms/step  step_ms %s" % (d['value'], d['ms_per_step'], d.get('step_ms')))
        cb = d.get('cpu_baseline') or {}
        if cb: print("cpu_baseline", {k: cb.get(k) for k in ('value', 'kind', 'cores', 's_per_step')})
        p = d.get('parity') or {}
        if p: print("parity", {k: p.get(k) for k in ('loss_rel', 'loss_rel_after_update', 'gradient_l2_rel', 'gradient_l2_rel_by_net', 'ok')})
        r = d.get('roofline') or {}
        if r: print("roofline", {k: v for k, v in r.items() if k not in ('by_kernel', 'by_call_group', 'timing', 'conv_family')}); print("conv_family", r.get('conv_family'))
        k = d.get('kernels') or {}
        if k: print("kernels", {n: (v.get('frac'), v.get('avg_us')) for n, v in k.items() if 'gbps' in v})
        if d.get('comm'): print("comm", json.dumps(d['comm'])[:600])
PY
}

for STEP in "$@"; do
  K=${STEP%%:*}; A=""; [ "$K" != "$STEP" ] && A=${STEP#*:}
  case $K in
  tests)
    if [ -n "$A" ]; then ( timeout 1200 python -m pytest tests -m gpu -q -x -k "$A" ) > $O/pytest_gpu_$TAG.log 2>&1
    else ( timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu_$TAG.log 2>&1; fi
    echo "pytest rc=$?"; grep -E "passed|failed|FAILED|^E  |Error" $O/pytest_gpu_$TAG.log | tail -12 ;;
  smoke)
    ( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke_$TAG.log ;;
  bench)
    ( timeout 700 python bench.py ) > $O/bench_$TAG.log 2> $O/bench_$TAG.err; echo "bench rc=$?"
    grep -E "cpu baseline|timed" $O/bench_$TAG.err | tail -4; line $O/bench_$TAG.log ;;
  quick)
    ( timeout 300 python bench.py $NOCPU --steps 30 ) > $O/bench_${TAG}_quick.log 2> $O/bench_${TAG}_quick.err; echo "quick rc=$?"
    grep timed $O/bench_${TAG}_quick.err ;;
  comm)
    ( CC_FORCE_COMM=1 timeout 300 python bench.py $NOCPU --steps 20 --warmup 5 ) > $O/bench_${TAG}_comm.log 2> $O/bench_${TAG}_comm.err; echo "comm rc=$?"
    line $O/bench_${TAG}_comm.log ;;
  cfg)
    for C in "--freeze" "--config c2" "--height 512 --width 1664 --batch 2" "--batch 8"; do
      F=$(echo "$C" | tr ' -' '__')
      ( timeout 400 python bench.py $NOCPU --steps 20 --warmup 5 $C ) > $O/bench_${TAG}_cfg$F.log 2> $O/bench_${TAG}_cfg$F.err
      echo "$C: $(grep timed $O/bench_${TAG}_cfg$F.err)"
    done ;;
  prof|prof0)
    # prof: the step as it runs (networks on their side streams: kernels overlap); prof0: CC_NET_STREAMS=0, kernels one after the other
    # (per-kernel durations that mean the kernel, the reference for roofline.avg_launch_us)
    PE=""; PT=$TAG; [ "$K" = prof0 ] && PE="CC_NET_STREAMS=0" && PT=${TAG}_serial
    ( cd /tmp && env $PE timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$PT -o $PT -- python $R/bench.py --steps 5 --warmup 2 $NOCPU ) > $O/rocprof_$PT.log 2>&1; echo "rocprof rc=$?"
    S=$(find $O/prof_$PT -name "*kernel_stats.csv" | head -1); T=$(find $O/prof_$PT -name "*kernel_trace.csv" | head -1)
    [ -n "$S" ] && cp "$S" $O/rocprof_kernel_stats_$PT.csv
    python tools/step_trace.py "$T" > $O/step_trace_$PT.txt 2>&1; head -${TRACE_ROWS:-45} $O/step_trace_$PT.txt
    find $O/prof_$PT -name "*kernel_trace.csv" -size +20M -delete ;;
  pmc)
    CMD="env CC_NET_STREAMS=0 python $R/bench.py --no-graph --steps 1 --warmup 1 $NOCPU"      # (per-kernel counters: kernels one after the other)
    for C in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pmc_$C
      ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -o run -- $CMD ) > $O/pmc_$C.log 2>&1; echo "pmc $C rc=$?"
    done
    F=$(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
    python tools/pmc_traffic.py "$F" "$W" $O/pmc_traffic.json "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python bench.py --no-graph --steps 1 --warmup 1" > $O/pmc_traffic_$TAG.txt; tail -6 $O/pmc_traffic_$TAG.txt
    cp $O/pmc_traffic.json profiles/pmc_traffic.json ;;
  calib)
    # FETCH_SIZE / WRITE_SIZE per access pattern on a known byte count (tools/fetch_calib.hip) -> profiles/fetch_calib.json
    [ -x tools/_bin/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/_bin/fetch_calib
    for C in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/cal_$C
      ( cd /tmp && timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/cal_$C -o run -- $R/tools/_bin/fetch_calib ) > $O/calib_$C.log 2>&1; echo "calib $C rc=$?"
    done
    F=$(find /tmp/cal_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find /tmp/cal_WRITE_SIZE -name "*counter_collection.csv" | head -1)
    python tools/fetch_calib.py "$F" "$W" $O/calib_FETCH_SIZE.log $O/fetch_calib.json | tee $O/fetch_calib_$TAG.txt
    cp $O/fetch_calib.json profiles/fetch_calib.json ;;
  sq)
    rm -rf /tmp/pmc_sq
    ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc_sq -o run -- env CC_NET_STREAMS=0 python $R/bench.py --no-graph --steps 1 --warmup 1 $NOCPU ) > $O/pmc_sq_$TAG.log 2>&1; echo "pmc sq rc=$?"
    F=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1)
    PMC_ROWS=${PMC_ROWS:-400} python tools/pmc_sq.py "$F" > $O/pmc_sq_$TAG.txt 2>&1; grep -E "^kernel|ssim|warp_|pose2flow" $O/pmc_sq_$TAG.txt | cut -c1-200 ;;
  mfma)
    # MFMA utilisation of the conv kernels (tools/pmc_mfma.py); needs the serial profile's kernel stats for the TFLOP/s column
    rm -rf /tmp/pmc_mfma
    ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma -o run -- env CC_NET_STREAMS=0 python $R/bench.py --no-graph --steps 1 --warmup 1 $NOCPU ) > $O/pmc_mfma_$TAG.log 2>&1; echo "pmc mfma rc=$?"
    F=$(find /tmp/pmc_mfma -name "*counter_collection.csv" | head -1)
    python tools/pmc_mfma.py "$F" $O/rocprof_kernel_stats_${TAG}_serial.csv > $O/pmc_mfma_$TAG.txt 2>&1; head -26 $O/pmc_mfma_$TAG.txt | cut -c1-120 ;;
  layers)
    # (the table comes from bench.py's isolated pass: side streams off)
    ( CC_TIMING_DETAIL=1 CC_TIMING_DUMP=$O/layers_$TAG.tsv timeout 400 python bench.py --steps 5 --warmup 3 --no-cpu-baseline ) > $O/bench_${TAG}_layers.log 2> $O/bench_${TAG}_layers.err
    python tools/layer_rates.py $O/layers_$TAG.tsv > $O/layer_rates_$TAG.txt; head -${LAYER_ROWS:-40} $O/layer_rates_$TAG.txt | cut -c1-170 ;;
  ab)
    export CC_LIB_PATH=${CC_LIB_PATH:-$TOOLS_LIB}
    IFS=';' read -ra VARS <<< "$A"
    for V in default "${VARS[@]}" default; do
      if [ "$V" = default ]; then E=""; else E=$(echo "$V" | tr ',' ' '); fi
      F=$(echo "$V" | tr ' =/,' '____')
      ( env $E timeout 300 python bench.py $NOCPU --steps 30 ) > $O/bench_${TAG}_$F.log 2> $O/bench_${TAG}_$F.err
      echo "$V: $(grep timed $O/bench_${TAG}_$F.err)"
    done
    unset CC_LIB_PATH ;;
  py)
    S=${A%%:*}; AR=""; [ "$S" != "$A" ] && AR=$(echo "${A#*:}" | tr ',' ' ')
    ( timeout 600 python tools/$S $AR ) > $O/${S%.py}_$TAG.txt 2>&1; echo "$S rc=$?"; tail -${PY_ROWS:-40} $O/${S%.py}_$TAG.txt ;;
  *) echo "unknown step $STEP" ;;
  esac
done
