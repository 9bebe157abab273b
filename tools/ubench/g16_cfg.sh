#!/bin/bash
# stand-alone time of the LDS-DMA bf16 kernel by configuration (diag build: XG_G16_CFG = <slab depth><ring stages>)
OUT=${1:-gpurun_out/g16}; mkdir -p $OUT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
: > $OUT/cfg.txt
for c in ${CFGS:-642 643 323 324 325}; do
  echo "== XG_G16_CFG=$c" >> $OUT/cfg.txt
  XG_G16_CFG=$c python tools/ubench/gemm16_bench.py both 2>/dev/null | cut -d'|' -f1 >> $OUT/cfg.txt
done
cat $OUT/cfg.txt
