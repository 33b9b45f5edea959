#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/gpu_r3u.sh 2>&1 | tail -5
bash tools/gpu_ab_env.sh r3s3 "CC_BIAS_CHUNK=4096" "CC_BIAS_CHUNK=16384" "CC_BIAS_SINGLE=1024" "CC_BIAS_SINGLE=16384" "CC_ACT_CHUNK=4096" "CC_ACT_CHUNK=16384"
