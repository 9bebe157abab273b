#!/bin/bash
# round-4 scheduling switches re-measured on the round-5 kernel and split rule (diag build)
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 2>/tmp/rs.err | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); r=d['roofline']; print('$1', d['ms_per_step'], r['in_situ_us_per_step'], d['final_loss'])
except Exception:
    print('$1 FAILED:', open('/tmp/rs.err').read()[-400:].replace(chr(10),' | '))
"; }
for i in 1 2; do
  run "default            :"
  XG_WG_BG=1 run "XG_WG_BG=1         :"
  XG_DEFER_WG=1 run "XG_DEFER_WG=1      :"
  XG_TOK_LATE=1 run "XG_TOK_LATE=1      :"
  XG_FWD_BG=0 run "XG_FWD_BG=0        :"
  for t in 8 9 11 12 13; do XG_BWD_TH=$t run "XG_BWD_TH=$t       :"; done
  for t in 8 9 11 12; do XG_FWD_TH=$t run "XG_FWD_TH=$t       :"; done
done
