#!/bin/bash
# where do the +0.8 ms of the data-parallel step form (1-rank RCCL group) come from?  one box:
#   plain | comm (two graphs) | comm, one graph + both all-reduces after it | comm without the per-step timing events
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { T=$1; shift; ( env "$@" timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 $EXTRA ) > gpurun_out/bench_r3y_$T.log 2> gpurun_out/bench_r3y_$T.err; echo "$T: $(grep timed gpurun_out/bench_r3y_$T.err)"; }
EXTRA="" run plain A=1
EXTRA="" run comm CC_FORCE_COMM=1
EXTRA="--split-graphs 0" run comm_onegraph CC_FORCE_COMM=1
EXTRA="" run comm_noevents CC_FORCE_COMM=1 CC_NO_COMM_EVENTS=1
EXTRA="--split-graphs 1" run plain_twographs A=1
EXTRA="" run plain2 A=1
