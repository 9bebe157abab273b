#!/bin/bash
# hidden-1024 bf16 iteration: split caps and step forms re-measured on the round-5 kernel (diag build)
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --workload xe5 --precision bf16 --steps 20 --warmup 6 2>/tmp/x5.err | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); r=d['roofline']; print('$1', d['ms_per_step'], r['in_situ_us_per_step'], d['final_loss'])
except Exception:
    print('$1 FAILED:', open('/tmp/x5.err').read()[-300:].replace(chr(10),' | '))
"; }
for i in 1 2; do
  run "default        :"
  XG_C1_KS=2 run "C1_KS=2        :"
  XG_C1_KS=8 run "C1_KS=8        :"
  XG_ENC_KS=1 run "ENC_KS=1       :"
  XG_ENC_KS=4 run "ENC_KS=4       :"
  XG_A_KS=1 run "A_KS=1         :"
  XG_A_KS=4 run "A_KS=4         :"
  XG_B_KS=1 run "B_KS=1         :"
  XG_B_KS=4 run "B_KS=4         :"
  XG_XE_FORM=D run "XE_FORM=D      :"
  XG_C1_LAG=4 run "C1_LAG=4       :"
  XG_C1_LAG=12 run "C1_LAG=12      :"
  XG_SK_DEEP_CHUNKS=32 run "DEEP_CHUNKS=32 :"
  XG_SK_DEEP_CHUNKS=128 run "DEEP_CHUNKS=128:"
done
