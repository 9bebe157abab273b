#!/bin/bash
# A/B of the LDS-DMA bf16 kernel (xg_gemm_g16.hip) against the register-staged one on the hidden-1024 shapes, then in the iteration.
# usage (GPU box): bash tools/ubench/g16_ab.sh [outdir]
OUT=${1:-gpurun_out/g16}; mkdir -p $OUT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
echo "== LDS-DMA kernel" > $OUT/gemm16.txt
python tools/ubench/gemm16_bench.py 2>/dev/null | cut -d'|' -f1 | sed 's/"fp32 operands".*"both bf16"/"both bf16"/' >> $OUT/gemm16.txt
echo "== register-staged kernel (XG_NO_G16=1)" >> $OUT/gemm16.txt
XG_NO_G16=1 python tools/ubench/gemm16_bench.py 2>/dev/null | cut -d'|' -f1 | sed 's/"fp32 operands".*"both bf16"/"both bf16"/' >> $OUT/gemm16.txt
cat $OUT/gemm16.txt
X5="python bench.py --workload xe5 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary"
for rep in 1 2; do
  XG_G16_NO8=1 $X5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xe5 bf16, LDS-DMA, no 8-wave form', d['ms_per_step'], d.get('parity_loss_delta'))" | tee -a $OUT/xe5.txt
  $X5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xe5 bf16, LDS-DMA kernel      ', d['ms_per_step'], d.get('parity_loss_delta'))" | tee -a $OUT/xe5.txt
  XG_NO_G16=1 $X5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xe5 bf16, register-staged     ', d['ms_per_step'], d.get('parity_loss_delta'))" | tee -a $OUT/xe5.txt
done
