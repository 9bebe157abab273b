#!/bin/bash
# hidden-1024 bf16 iteration with the skinny launches forced to 8-wave / 4-wave workgroups (diag build)
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --workload xe5 --precision bf16 --steps 20 --warmup 6 2>/tmp/nw.err | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); r=d['roofline']; print('$1', d['ms_per_step'], r['avg_launch_us'], r['in_situ_us_per_step'], d['final_loss'])
except Exception:
    print('$1 FAILED:', open('/tmp/nw.err').read()[-400:].replace(chr(10),' | '))
"; }
for i in 1 2; do
  run "default rule :"
  XG_SK_NW=8 run "XG_SK_NW=8   :"
  XG_SK_NW=4 run "XG_SK_NW=4   :"
done
