#!/bin/bash
# HIP runtime knobs on the plain (single-GPU) step: kernel-argument placement, graph packet capture
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { T=$1; shift; ( env "$@" timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 40 ) > gpurun_out/bench_r3k2_$T.log 2> gpurun_out/bench_r3k2_$T.err; echo "$T: $(grep timed gpurun_out/bench_r3k2_$T.err)"; }
run plain A=1
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run pktcapture1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run pktcapture0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run plain2 A=1
