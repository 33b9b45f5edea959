#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/gpu_ab_env.sh r3s3 "CC_CONV_BALANCE=0" "CC_W3_MINM=65 CC_WGRAD_THIN_MAXCOMBO=16" "CC_NO_WGRAD_QUEUE=1" "CC_CONV_BALANCE_PCT=90"
