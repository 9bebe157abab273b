#!/bin/bash
# split-bf16 iteration with pre-split weight planes (product library) vs exact fp32, same box, alternating
OUT=${1:-gpurun_out/g16}; mkdir -p $OUT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], "step group us:", d["roofline"].get("avg_launch_us"), "in situ:", d["roofline"].get("in_situ_us_per_step"), "parity", d.get("parity_loss_delta"))'
for rep in 1 2 3; do
  $B --precision bf16x3 2>/dev/null | python -c "$P" "x3 pre-split planes" | tee -a $OUT/x3_planes.txt
  XG_X3_TILES=0 $B --precision bf16x3 2>/dev/null | python -c "$P" "x3 fp32 tiles      " | tee -a $OUT/x3_planes.txt
  $B 2>/dev/null | python -c "$P" "exact fp32         " | tee -a $OUT/x3_planes.txt
done
$B --precision bf16x3 --workload scst 2>/dev/null | python -c "$P" "scst x3" | tee -a $OUT/x3_planes.txt
$B --workload scst 2>/dev/null | python -c "$P" "scst fp32" | tee -a $OUT/x3_planes.txt
