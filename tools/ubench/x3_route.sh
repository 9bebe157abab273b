#!/bin/bash
# split-bf16 iteration with classes of products routed to exact fp32 (diag build: XG_X3_FP32 mask 1 TN, 2 NT, 4 NN)
OUT=${1:-gpurun_out/g16}; mkdir -p $OUT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d.get("parity_loss_delta"))'
for rep in 1 2; do for m in ${MASKS:-0 1 3 5}; do
  XG_X3_FP32=$m $B --precision bf16x3 2>/dev/null | python -c "$P" "xe bf16x3 XG_X3_FP32=$m " | tee -a $OUT/x3_route.txt
done
$B 2>/dev/null | python -c "$P" "xe fp32   " | tee -a $OUT/x3_route.txt
done
