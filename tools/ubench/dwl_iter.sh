#!/bin/bash
# the iteration with the vocabulary head's weight gradient enqueued from inside the reverse-time loop (diag library, XG_DWL_AT=<step>)
cd $GRAFT_REPO_ROOT
run() { XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['ms_per_step'], r['avg_launch_us'], r['in_situ_us_per_step'], d['final_loss'])"; }
for i in 1 2; do run "default :"; for at in ${DWL_STEPS:-20 12 10 9 6}; do XG_DWL_AT=$at run "XG_DWL_AT=$at :"; done; done
