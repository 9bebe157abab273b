#!/bin/bash
# encoder weight gradients: later half of the frames under the backward recurrence (default) vs all frames behind it (diag: XG_ENC_WG_WHOLE=1)
OUT=${1:-gpurun_out/g16}; mkdir -p $OUT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d.get("parity_loss_delta"))'
for rep in 1 2 3; do
  $B 2>/dev/null | python -c "$P" "xe fp32   split" | tee -a $OUT/encwg.txt
  XG_ENC_WG_WHOLE=1 $B 2>/dev/null | python -c "$P" "xe fp32   whole" | tee -a $OUT/encwg.txt
done
for rep in 1 2; do
  $B --workload xe5 --precision bf16 2>/dev/null | python -c "$P" "xe5 bf16  split" | tee -a $OUT/encwg.txt
  XG_ENC_WG_WHOLE=1 $B --workload xe5 --precision bf16 2>/dev/null | python -c "$P" "xe5 bf16  whole" | tee -a $OUT/encwg.txt
  $B --workload scst 2>/dev/null | python -c "$P" "scst fp32 split" | tee -a $OUT/encwg.txt
  XG_ENC_WG_WHOLE=1 $B --workload scst 2>/dev/null | python -c "$P" "scst fp32 whole" | tee -a $OUT/encwg.txt
  $B --precision bf16x3 2>/dev/null | python -c "$P" "xe bf16x3 split" | tee -a $OUT/encwg.txt
  XG_ENC_WG_WHOLE=1 $B --precision bf16x3 2>/dev/null | python -c "$P" "xe bf16x3 whole" | tee -a $OUT/encwg.txt
done
