#!/bin/bash
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 $2 2>/tmp/sn.err | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); r=d['roofline']; print('$1', d['ms_per_step'], r['avg_launch_us'], r['in_situ_us_per_step'], d['final_loss'])
except Exception:
    print('$1 FAILED:', open('/tmp/sn.err').read()[-400:].replace(chr(10),' | '))
"; }
for i in 1 2 3; do
  run "default (8-wave unsplit launches) :" ""
  XG_SK_NW=4 run "XG_SK_NW=4                        :" ""
done
