#!/bin/bash
# upper bound of pre-split weight planes for the split-bf16 per-step products: a build whose weight-side split is free
# (lib/libxgate_hip_splitfree.so = -DSKF_SPLIT_FREE, built by __graft_entry__.build_variant; results wrong, timing only)
OUT=${1:-gpurun_out/g16}; mkdir -p $OUT
V=$PWD/controllable_xgating_amd/lib/libxgate_hip_splitfree.so
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary --precision bf16x3"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], "step group us:", d["roofline"].get("avg_launch_us"), "in situ:", d["roofline"].get("in_situ_us_per_step"))'
for rep in 1 2; do
  $B 2>/dev/null | python -c "$P" "x3 product lib      " | tee -a $OUT/split_free.txt
  XG_LIBRARY=$V $B 2>/dev/null | python -c "$P" "x3 free weight split" | tee -a $OUT/split_free.txt
done
