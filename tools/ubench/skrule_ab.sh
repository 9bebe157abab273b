#!/bin/bash
# products alone in the three arithmetic modes + the iterations that use the register-staged bf16 kernels
OUT=${1:-gpurun_out/g16}; mkdir -p $OUT
python tools/ubench/gemm_bench.py 2>/dev/null | tee $OUT/gemm_bench_raw.txt | cut -c1-1500
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d.get("parity_loss_delta"))'
for rep in 1 2; do
  $B --precision bf16x3 2>/dev/null | python -c "$P" "xe bf16x3 " | tee -a $OUT/skrule.txt
  $B 2>/dev/null | python -c "$P" "xe fp32   " | tee -a $OUT/skrule.txt
  $B --workload xe5 --precision bf16 2>/dev/null | python -c "$P" "xe5 bf16  " | tee -a $OUT/skrule.txt
  $B --precision bf16 2>/dev/null | python -c "$P" "xe bf16   " | tee -a $OUT/skrule.txt
done
