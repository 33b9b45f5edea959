#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/aten_sites.py gpurun_out/aten_sites_r3r.txt 2> gpurun_out/aten_sites_r3r.err; echo rc=$?
head -120 gpurun_out/aten_sites_r3r.txt; tail -5 gpurun_out/aten_sites_r3r.err
