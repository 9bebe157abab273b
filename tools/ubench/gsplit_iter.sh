#!/bin/bash
# split target of the 64 x 64-tile products inside the iteration (diag build, XG_GEMM_SPLIT_PCT)
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 2>/tmp/gs.err | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); r=d['roofline']; print('$1', d['ms_per_step'], r['in_situ_us_per_step'], d['final_loss'])
except Exception:
    print('$1 FAILED:', open('/tmp/gs.err').read()[-300:].replace(chr(10),' | '))
"; }
for i in 1 2 3; do
  for p in 100 25 50 75 150 200; do XG_GEMM_SPLIT_PCT=$p run "split target $p % :"; done
done
