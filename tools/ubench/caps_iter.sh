#!/bin/bash
# per-launch split caps under the round-5 split rule (diag build)
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 2>/tmp/cp.err | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); r=d['roofline']; print('$1', d['ms_per_step'], r['in_situ_us_per_step'], d['final_loss'])
except Exception:
    print('$1 FAILED:', open('/tmp/cp.err').read()[-400:].replace(chr(10),' | '))
"; }
for i in 1 2 3; do
  run "new rule (enc 2, A 2, B 4, c1 2, lag 7) :"
  XG_C1_KS=4 run "c1 4                                    :"
  XG_C1_KS=4 XG_C1_LAG=10 run "c1 4 lag 10                             :"
  XG_C1_KS=8 run "c1 8                                    :"
  XG_C1_KS=4 XG_C1_LAG=4 run "c1 4 lag 4                              :"
  XG_C1_KS=4 XG_C1_LOWPRIO=0 run "c1 4 default priority                   :"
  XG_C1_KS=4 XG_WG_CHUNKS=3 run "c1 4, weight gradients in 3 chunks      :"
  XG_C1_KS=4 XG_WG_CHUNKS=1 run "c1 4, weight gradients in 1 chunk       :"
done
