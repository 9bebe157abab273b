#!/bin/bash
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$GRAFT_REPO_ROOT/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { local name=$1; shift; local wl=$1; shift
  local out=$(env "$@" timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --workload $wl --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('in_situ_us_per_step'), d.get('final_loss'))")
  echo "$name $wl : $out"; }
run no_td xe XG_GEMM_NO_TD=1
run td_onlybg xe XG_TD_ONLY_BG=1
run td_pad xe XG_TD_PAD=1
run td_pad_256 xe XG_TD_PAD=1 XG_TD_TILES_MIN=200
run no_td xe XG_GEMM_NO_TD=1
run td_onlybg xe XG_TD_ONLY_BG=1
run td_pad xe XG_TD_PAD=1
run td_onlybg scst XG_TD_ONLY_BG=1
run no_td scst XG_GEMM_NO_TD=1
