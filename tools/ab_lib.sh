#!/bin/bash
# Same-box A/B of LIBRARY BUILDS (kernel-source variants compiled with -D switches into tools/_bin/ before the gpurun call):
#     gpurun -- 'bash tools/ab_lib.sh TAG "ENV" lib1.so lib2.so ...'      ("-" = the product library; ENV e.g. CC_NET_STREAMS=0 or "")
# 30-step bench per library, the first one run again at the end.
TAG=$1; ENVV=$2; shift 2
mkdir -p gpurun_out
FIRST=$1
for L in "$@" "$FIRST"; do
  if [ "$L" = "-" ]; then P=""; N=product; else P="CC_LIB_PATH=$PWD/tools/_bin/$L"; N=${L%.so}; fi
  ( env $ENVV $P timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 ) > gpurun_out/ablib_${TAG}_$N.log 2> gpurun_out/ablib_${TAG}_$N.err
  echo "$N $ENVV: $(grep timed gpurun_out/ablib_${TAG}_$N.err)"
done
