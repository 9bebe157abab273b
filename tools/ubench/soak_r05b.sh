#!/bin/bash
# soak of the configurations whose kernels changed in the last session of round 5 (one fixed synthetic batch)
OUT=${1:-gpurun_out/g16}; mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-pmc --no-secondary --warmup 5"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "steps", d["steps"], "ms", d["ms_per_step"], "final loss", d.get("final_loss"), "parity", d.get("parity_loss_delta"))'
$B --workload xe5 --precision bf16 --steps 1000 2>/dev/null | python -c "$P" "xe5 bf16 " | tee -a $OUT/soak.txt
$B --precision bf16x3 --steps 2000 2>/dev/null | python -c "$P" "xe bf16x3" | tee -a $OUT/soak.txt
$B --precision bf16 --steps 1000 2>/dev/null | python -c "$P" "xe bf16  " | tee -a $OUT/soak.txt
$B --steps 2000 2>/dev/null | python -c "$P" "xe fp32  " | tee -a $OUT/soak.txt
