#!/bin/bash
# LDS-DMA products launched beside a chain kept to one workgroup per CU (diag: XG_G16_BG=1) vs two
OUT=${1:-gpurun_out/g16}; mkdir -p $OUT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
X5="python bench.py --workload xe5 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], "in situ step", d["roofline"].get("in_situ_us_per_step"))'
for rep in 1 2 3; do
  $X5 2>/dev/null | python -c "$P" "xe5 default      " | tee -a $OUT/g16_bg.txt
  XG_G16_BG=1 $X5 2>/dev/null | python -c "$P" "xe5 bg = 1 per CU" | tee -a $OUT/g16_bg.txt
done
