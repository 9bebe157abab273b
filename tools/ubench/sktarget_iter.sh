#!/bin/bash
# how many workgroups a cross-workgroup split launch may fill (diag build, XG_SK_TARGET; 512 = the round-4 rule)
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 $2 2>/tmp/st.err | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); r=d['roofline']; print('$1', d['ms_per_step'], r['in_situ_us_per_step'], d['final_loss'])
except Exception:
    print('$1 FAILED:', open('/tmp/st.err').read()[-400:].replace(chr(10),' | '))
"; }
for i in 1 2 3; do
  XG_SK_TARGET=512 run "xe      round-4 rule (512)      :" ""
  run "xe      new rule                :" ""
  XG_SK_TARGET=512 run "xe5bf16 round-4 rule (512)      :" "--workload xe5 --precision bf16"
  run "xe5bf16 new rule                :" "--workload xe5 --precision bf16"
  XG_SK_DEEP_CHUNKS=24 run "xe5bf16 new rule, deep >= 24    :" "--workload xe5 --precision bf16"
  XG_SK_TARGET=512 run "scst    round-4 rule (512)      :" "--workload scst"
  run "scst    new rule                :" "--workload scst"
done
