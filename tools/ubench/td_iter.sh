#!/bin/bash
# the iteration with / without gemm_td_kernel (diag library), XE fp32 and SCST
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$GRAFT_REPO_ROOT/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { local name=$1; shift; local wl=$1; shift
  local out=$(env "$@" timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --workload $wl --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('in_situ_us_per_step'), d.get('final_loss'))")
  echo "$name $wl : $out"; }
run td xe A=1
run no_td xe XG_GEMM_NO_TD=1
run td xe A=1
run no_td xe XG_GEMM_NO_TD=1
run td scst A=1
run no_td scst XG_GEMM_NO_TD=1
echo "== bg form of the products alone"
XG_GEMM_SHAPES="TN" XG_GEMM_FORCE_BG=1 python tools/ubench/gemm_bench.py one 0 2>/dev/null | tail -1 | cut -c1-200
XG_GEMM_SHAPES="TN" XG_GEMM_FORCE_BG=1 XG_GEMM_NO_TD=1 python tools/ubench/gemm_bench.py one 0 2>/dev/null | tail -1 | cut -c1-200
