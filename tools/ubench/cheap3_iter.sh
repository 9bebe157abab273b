#!/bin/bash
# upper bound for split-bf16 products with PRE-SPLIT planes: the -DXG_SPLIT_CHEAP build stores one truncated plane three times
# (results wrong, timing only); compared with the real split and with exact fp32
cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 $2 2>/tmp/c3.err | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); r=d['roofline']; print('$1', d['ms_per_step'], r['in_situ_us_per_step'], d['final_loss'])
except Exception:
    print('$1 FAILED:', open('/tmp/c3.err').read()[-400:].replace(chr(10),' | '))
"; }
for i in 1 2; do
  run "fp32 exact            :" ""
  run "split-bf16 (real)     :" "--precision bf16x3"
  XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_cheap3.so run "split-bf16 (free split):" "--precision bf16x3"
done
echo "== stand-alone products, mode 3: real split, then free split"
python tools/ubench/gemm_bench.py 2>&1 | grep "mode 3"
XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_cheap3.so python tools/ubench/gemm_bench.py 2>&1 | grep "mode 3"
