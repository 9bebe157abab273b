#!/bin/bash
# plain products through the vendor library (diag build, XG_BLASLT=<mask>): stand-alone shapes, then the iteration
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
echo "== stand-alone (tools/ubench/gemm_bench.py mode 0): own kernels, then XG_BLASLT=39 (every class incl. bias-gradient products)"
python tools/ubench/gemm_bench.py 2>&1 | grep "mode 0"
XG_BLASLT=39 python tools/ubench/gemm_bench.py 2>&1 | grep "mode 0"
run() { timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 2>/tmp/bl.err | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); r=d['roofline']; print('$1', d['ms_per_step'], r['in_situ_us_per_step'], d['final_loss'])
except Exception:
    print('$1 FAILED:', open('/tmp/bl.err').read()[-600:].replace(chr(10),' | '))
"; }
echo "== iteration"
for i in 1 2; do
  run "own kernels :"
  for m in ${LT_MASKS:-2 4 1 33 10 18 20 17 49 7 39 103}; do XG_BLASLT=$m run "XG_BLASLT=$m :"; done
done
