#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/gpu_ab_env.sh r3s3 "CC_CONV_TW16_TIES=1" "A=1" "CC_CONV_TW16_TIES=1"
