#!/bin/bash
# split-K of the LDS-DMA kernel forced (diag build: XG_G16_SK), per configuration
OUT=${1:-gpurun_out/g16}; mkdir -p $OUT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
: > $OUT/sk.txt
for c in ${CFGS:-323 642}; do for k in ${SKS:-1 2 3 4}; do
  echo "== XG_G16_CFG=$c XG_G16_SK=$k" >> $OUT/sk.txt
  XG_G16_CFG=$c XG_G16_SK=$k python tools/ubench/gemm16_bench.py both 2>/dev/null | cut -d'|' -f1 | sed 's/err.*//' >> $OUT/sk.txt
done; done
cat $OUT/sk.txt
