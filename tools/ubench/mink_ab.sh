#!/bin/bash
# shallow reductions (K < 256) on the fp32 kernels (default) vs on the bf16 tile kernels (diag: XG_BF16_MINK=64) in modes 1 / 3
OUT=${1:-gpurun_out/g16}; mkdir -p $OUT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d.get("parity_loss_delta"))'
for rep in 1 2; do
  $B --workload xe5 --precision bf16 2>/dev/null | python -c "$P" "xe5 bf16  K>=256" | tee -a $OUT/mink.txt
  XG_BF16_MINK=64 $B --workload xe5 --precision bf16 2>/dev/null | python -c "$P" "xe5 bf16  K>=64 " | tee -a $OUT/mink.txt
  $B --precision bf16x3 2>/dev/null | python -c "$P" "xe bf16x3 K>=256" | tee -a $OUT/mink.txt
  XG_BF16_MINK=64 $B --precision bf16x3 2>/dev/null | python -c "$P" "xe bf16x3 K>=64 " | tee -a $OUT/mink.txt
done
