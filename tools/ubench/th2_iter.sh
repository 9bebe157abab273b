#!/bin/bash
# forward: a second background vocabulary product (diag build, XG_FWD_TH=<a> XG_FWD_TH2=<b>): steps [0,a) at step a, [a,b) at step b, the rest behind the loop
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 2>/tmp/t2.err | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); r=d['roofline']; print('$1', d['ms_per_step'], r['in_situ_us_per_step'], d['final_loss'])
except Exception:
    print('$1 FAILED:', open('/tmp/t2.err').read()[-400:].replace(chr(10),' | '))
"; }
for i in 1 2; do
  run "default (10 | 11 behind)      :"
  XG_FWD_TH=7 XG_FWD_TH2=14 run "7 | 7 | 7 behind             :"
  XG_FWD_TH=8 XG_FWD_TH2=15 run "8 | 7 | 6 behind             :"
  XG_FWD_TH=9 XG_FWD_TH2=16 run "9 | 7 | 5 behind             :"
  XG_FWD_TH=10 XG_FWD_TH2=16 run "10 | 6 | 5 behind            :"
  XG_FWD_TH=10 XG_FWD_TH2=17 run "10 | 7 | 4 behind            :"
done
