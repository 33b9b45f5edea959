#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/gpu_ab_env.sh r3s3 "CC_CONV_BM_PADSAVE=20" "CC_CONV_BM_PADSAVE=25" "CC_CONV_BM_PADSAVE=35" "CC_CONV_BM_PADSAVE=12"
