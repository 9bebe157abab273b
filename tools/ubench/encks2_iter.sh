#!/bin/bash
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 2>/tmp/ek.err | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); r=d['roofline']; print('$1', d['ms_per_step'], r['in_situ_us_per_step'], d['final_loss'])
except Exception:
    print('$1 FAILED:', open('/tmp/ek.err').read()[-400:].replace(chr(10),' | '))
"; }
for i in 1 2 3; do
  run "default                 :"
  XG_ENC_KS=2 run "ENC_KS=2                :"
  XG_ENC_KS=3 run "ENC_KS=3                :"
  XG_ENC_KS=2 XG_A_KS=2 run "ENC_KS=2 A_KS=2         :"
  XG_ENC_KS=2 XG_B_KS=2 run "ENC_KS=2 B_KS=2         :"
  XG_ENC_KS=2 XG_A_KS=4 XG_B_KS=4 run "ENC_KS=2 A_KS=4 B_KS=4  :"
done
