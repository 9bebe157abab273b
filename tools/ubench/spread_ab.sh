#!/bin/bash
# per-step kernel: the next chunk's operand requests between the MFMA groups (lib/libxgate_hip_spread.so = -DSKF_LOADS_SPREAD) vs in one burst (product)
OUT=${1:-gpurun_out/g16}; mkdir -p $OUT
V=$PWD/controllable_xgating_amd/lib/libxgate_hip_spread.so
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], "step group us:", d["roofline"].get("avg_launch_us"), "in situ:", d["roofline"].get("in_situ_us_per_step"), "parity", d.get("parity_loss_delta"))'
for rep in 1 2 3; do
  $B 2>/dev/null | python -c "$P" "fp32 burst (product)" | tee -a $OUT/spread.txt
  XG_LIBRARY=$V $B 2>/dev/null | python -c "$P" "fp32 spread         " | tee -a $OUT/spread.txt
done
$B --precision bf16x3 2>/dev/null | python -c "$P" "x3 burst" | tee -a $OUT/spread.txt
XG_LIBRARY=$V $B --precision bf16x3 2>/dev/null | python -c "$P" "x3 spread" | tee -a $OUT/spread.txt
$B --precision bf16 --workload xe5 2>/dev/null | python -c "$P" "xe5 burst" | tee -a $OUT/spread.txt
XG_LIBRARY=$V $B --precision bf16 --workload xe5 2>/dev/null | python -c "$P" "xe5 spread" | tee -a $OUT/spread.txt
