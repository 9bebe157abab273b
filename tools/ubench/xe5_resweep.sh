#!/bin/bash
# hidden-1024 bf16 iteration: scheduling switches of the per-step launches re-measured beside the LDS-DMA product kernel (diag build)
OUT=${1:-gpurun_out/g16}; mkdir -p $OUT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
X5="python bench.py --workload xe5 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-secondary"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"])'
run() { env "$@" $X5 2>/dev/null | python -c "$P" "$*" | tee -a $OUT/xe5_resweep.txt; }
run XG_DUMMY=1
run XG_SK_NW=8
run XG_SK_SPLIT_NW=8
run XG_SK_TARGET=512
run XG_SK_TARGET=128
run XG_C1_KS=2
run XG_C1_LAG=3
run XG_C1_LAG=21
run XG_XE_FORM=E
run XG_WG_CHUNKS=1
run XG_DEFER_WG=1
run XG_DUMMY=2
