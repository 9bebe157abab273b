#!/bin/bash
# kernel-routing switches re-measured on the final round-5 state (diag build)
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 2>/tmp/rs.err | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); r=d['roofline']; print('$1', d['ms_per_step'], r['in_situ_us_per_step'], d['final_loss'])
except Exception:
    print('$1 FAILED:', open('/tmp/rs.err').read()[-300:].replace(chr(10),' | '))
"; }
for i in 1 2; do
  run "default              :"
  XG_ATTN_BWD_ONE=1 run "XG_ATTN_BWD_ONE=1    :"
  XG_GEMM_NO_BG=1 run "XG_GEMM_NO_BG=1      :"
  XG_GEMM_NO_TD=1 run "XG_GEMM_NO_TD=1      :"
  XG_GEMM_NO_W1=1 run "XG_GEMM_NO_W1=1      :"
  XG_GEMM_NO_PK=1 run "XG_GEMM_NO_PK=1      :"
  XG_TD_ALL=1 run "XG_TD_ALL=1          :"
  XG_S1_FIRST=0 run "XG_S1_FIRST=0        :"
  XG_S1_FIRST=1 run "XG_S1_FIRST=1        :"
  XG_L1_ORDER=0 run "XG_L1_ORDER=0        :"
  XG_C1_LAG=21 run "XG_C1_LAG=21         :"
  XG_C1_LAG=3 run "XG_C1_LAG=3          :"
done
