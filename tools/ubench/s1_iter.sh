#!/bin/bash
# rollout form of the step: S1' (cell 1's recurrent + token products) in launch 1 (XG_S1_FIRST=1) -- stand-alone step group and the SCST iteration
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 20 --warmup 6 $2 2>/tmp/s1.err | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); r=d['roofline']; print('$1', d['ms_per_step'], 'step group', r['avg_launch_us'], 'in situ', r['in_situ_us_per_step'], r.get('step_us_by_arithmetic'))
except Exception:
    print('$1 FAILED:', open('/tmp/s1.err').read()[-300:].replace(chr(10),' | '))
"; }
for i in 1 2 3; do
  run "scst default        :" "--workload scst"
  XG_S1_FIRST=1 run "scst XG_S1_FIRST=1  :" "--workload scst"
  XG_S1_FIRST=1 XG_L1_ORDER=0 run "scst S1_FIRST, L1_ORDER=0 :" "--workload scst"
done
