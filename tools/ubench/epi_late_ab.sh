#!/bin/bash
# per-step kernel: epilogue operand requests behind the K loop's first operand request (lib/libxgate_hip_epilate.so = -DSKF_EPI_LATE) vs in front of it (product)
OUT=${1:-gpurun_out/g16}; mkdir -p $OUT
V=$PWD/controllable_xgating_amd/lib/libxgate_hip_epilate.so
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], "step group us:", d["roofline"].get("avg_launch_us"), "in situ:", d["roofline"].get("in_situ_us_per_step"), "parity", d.get("parity_loss_delta"))'
for rep in 1 2 3; do
  $B 2>/dev/null | python -c "$P" "fp32 product " | tee -a $OUT/epi_late.txt
  XG_LIBRARY=$V $B 2>/dev/null | python -c "$P" "fp32 epi late" | tee -a $OUT/epi_late.txt
done
$B --workload scst 2>/dev/null | python -c "$P" "scst product " | tee -a $OUT/epi_late.txt
XG_LIBRARY=$V $B --workload scst 2>/dev/null | python -c "$P" "scst epi late" | tee -a $OUT/epi_late.txt
