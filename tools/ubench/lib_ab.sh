#!/bin/bash
# product library vs diag library (defaults) on the same box
B="python bench.py --no-cpu-baseline --no-pmc --no-secondary --steps 20 --warmup 5"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"])'
D=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
for rep in 1 2; do
$B --precision bf16x3 2>/dev/null | python -c "$P" "x3 product lib"
XG_LIBRARY=$D $B --precision bf16x3 2>/dev/null | python -c "$P" "x3 diag lib"
XG_LIBRARY=$D XG_X3_FP32=0 $B --precision bf16x3 2>/dev/null | python -c "$P" "x3 diag lib, mask 0"
$B 2>/dev/null | python -c "$P" "fp32 product lib"
XG_LIBRARY=$D $B 2>/dev/null | python -c "$P" "fp32 diag lib"
done
