#!/bin/bash
# the iteration with the library's side streams restricted to the first n CUs of every XCD (diag library, XG_AUX_CUMASK=n[,m])
cd $GRAFT_REPO_ROOT
run() { XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['ms_per_step'], r['avg_launch_us'], r['in_situ_us_per_step'])"; }
for m in ${MASKS:-0 28 24 20 16 24,32 32,24 0}; do XG_AUX_CUMASK=$m run "mask $m :"; done
