#!/bin/bash
# split launches of at most one workgroup per CU as 8-wave workgroups (diag build, XG_SK_SPLIT_NW=8)
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 $2 2>/tmp/sn.err | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); r=d['roofline']; print('$1', d['ms_per_step'], r['in_situ_us_per_step'], d['final_loss'], d.get('parity_loss_delta'))
except Exception:
    print('$1 FAILED:', open('/tmp/sn.err').read()[-400:].replace(chr(10),' | '))
"; }
for i in 1 2 3; do
  run "4-wave split launches :" ""
  XG_SK_SPLIT_NW=8 run "8-wave split launches :" ""
done
