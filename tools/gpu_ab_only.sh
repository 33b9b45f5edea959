#!/bin/bash
# same-box A/B only: bash tools/gpu_ab_only.sh TAG VAR=VAL ...
TAG=$1; shift
bash tools/gpu_ab_env.sh $TAG "$@"
