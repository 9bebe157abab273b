#!/bin/bash
# split cap of the register-staged bf16 / split-bf16 kernel (diag build: XG_BS_SKMAX): products alone (mode 3 and mode 1 at hidden 512)
OUT=${1:-gpurun_out/g16}; mkdir -p $OUT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
export XG_GEMM_SHAPES="wgrad,enc embed,PRE,vproj,dX,dH NN K,dW_logit"
: > $OUT/skmax.txt
for k in ${SKS:-99 4 3 2 1}; do
  echo "== XG_BS_SKMAX=$k" >> $OUT/skmax.txt
  XG_BS_SKMAX=$k python tools/ubench/gemm_bench.py one 3 2>/dev/null >> $OUT/skmax.txt
done
cat $OUT/skmax.txt
