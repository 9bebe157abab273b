#!/bin/bash
# the iteration with the library's side streams at the lowest queue priority (diag library, XG_AUX_PRIO=1)
cd $GRAFT_REPO_ROOT
run() { XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['ms_per_step'], r['avg_launch_us'], r['in_situ_us_per_step'])"; }
for i in 1 2 3; do run "default :"; XG_AUX_PRIO=1 run "aux low priority :"; done
