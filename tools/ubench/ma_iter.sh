#!/bin/bash
# reverse-time loop in its two-launch form (dalpha from ds2 . M) against the three-launch form (diag build, XG_MA=1)
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 $2 2>/tmp/ma.err | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); r=d['roofline']; print('$1', d['ms_per_step'], r['in_situ_us_per_step'], d['final_loss'])
except Exception:
    print('$1 FAILED:', open('/tmp/ma.err').read()[-400:].replace(chr(10),' | '))
"; }
for i in 1 2 3; do
  XG_MA=1 run "two-launch form (XG_MA=1)   :" "$MA_ARGS"
  run "three-launch form (default):" "$MA_ARGS"
done
