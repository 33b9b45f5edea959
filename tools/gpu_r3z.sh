#!/bin/bash
# AMD_DIRECT_DISPATCH=0 (HIP runtime submits through a per-queue worker thread: a stream-wait on another stream's event does not
# block the calling thread): plain and data-parallel step forms, same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { T=$1; shift; ( env "$@" timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 $EXTRA ) > gpurun_out/bench_r3z_$T.log 2> gpurun_out/bench_r3z_$T.err; echo "$T: $(grep timed gpurun_out/bench_r3z_$T.err)"; }
EXTRA="" run plain A=1
EXTRA="" run plain_nodirect AMD_DIRECT_DISPATCH=0
EXTRA="" run comm CC_FORCE_COMM=1
EXTRA="" run comm_nodirect CC_FORCE_COMM=1 AMD_DIRECT_DISPATCH=0
EXTRA="--freeze" run freeze_comm CC_FORCE_COMM=1
EXTRA="--freeze" run freeze_comm_nodirect CC_FORCE_COMM=1 AMD_DIRECT_DISPATCH=0
EXTRA="" run plain2 A=1
EXTRA="" run plain_nodirect2 AMD_DIRECT_DISPATCH=0
( AMD_DIRECT_DISPATCH=0 timeout 600 python -m pytest tests/test_nets_gpu.py -m gpu -q -x ) > gpurun_out/pytest_r3z.log 2>&1; echo "pytest(nets, AMD_DIRECT_DISPATCH=0) rc=$?"; tail -1 gpurun_out/pytest_r3z.log
AMD_DIRECT_DISPATCH=0 timeout 300 python tools/host_probe.py 2>/dev/null | tail -2
