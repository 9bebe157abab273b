#!/bin/bash
# DMA instructions between the MFMA groups (default) vs in one burst in front of the fragment reads (diag: XG_G16_DBG=8)
OUT=${1:-gpurun_out/g16}; mkdir -p $OUT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
: > $OUT/burst.txt
for c in ${CFGS:-642 323 844}; do for d in 0 8; do
  echo "== XG_G16_CFG=$c XG_G16_DBG=$d" >> $OUT/burst.txt
  XG_G16_CFG=$c XG_G16_DBG=$d python tools/ubench/gemm16_bench.py both 2>/dev/null | cut -d'|' -f1 | sed 's/err.*//' >> $OUT/burst.txt
done; done
cat $OUT/burst.txt
