#!/bin/bash
# timing-only ablations of the LDS-DMA kernel (diag build; results wrong): XG_G16_DBG bit 0 = no DMA inside the loop,
# bit 1 = no fragment reads / MFMAs, bit 2 = no epilogue
OUT=${1:-gpurun_out/g16}; mkdir -p $OUT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
: > $OUT/dbg.txt
for c in ${CFGS:-323}; do for d in 0 1 2 4 5 6; do
  echo "== XG_G16_CFG=$c XG_G16_DBG=$d" >> $OUT/dbg.txt
  XG_G16_CFG=$c XG_G16_DBG=$d python tools/ubench/gemm16_bench.py both 2>/dev/null | cut -d'|' -f1 | sed 's/err.*//' >> $OUT/dbg.txt
done; done
cat $OUT/dbg.txt
