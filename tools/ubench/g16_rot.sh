#!/bin/bash
# tile-dependent start of the slab walk (default) vs every workgroup from the first slab (diag: XG_G16_DBG=16)
OUT=${1:-gpurun_out/g16}; mkdir -p $OUT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
: > $OUT/rot.txt
for c in ${CFGS:-642 323 844}; do for d in 0 16; do
  echo "== XG_G16_CFG=$c XG_G16_DBG=$d" >> $OUT/rot.txt
  XG_G16_CFG=$c XG_G16_DBG=$d python tools/ubench/gemm16_bench.py both 2>/dev/null | cut -d'|' -f1 | sed 's/err.*//' >> $OUT/rot.txt
done; done
cat $OUT/rot.txt
