#!/bin/bash
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$GRAFT_REPO_ROOT/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { local name=$1; shift; local wl=$1; shift
  local out=$(env "$@" timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --workload $wl --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('in_situ_us_per_step'), d.get('final_loss'))")
  echo "$name $wl : $out"; }
run no_td xe XG_GEMM_NO_TD=1
run only_big xe XG_TD_TILES_MIN=1024
run only_small xe XG_TD_TILES_MAX=1024
run only_256 xe XG_TD_TILES_MIN=200 XG_TD_TILES_MAX=300
run lt200 xe XG_TD_TILES_MAX=199
run no_td xe XG_GEMM_NO_TD=1
run td xe A=1
run td_d4 xe XG_TD_DEPTH=4
run td_bgd4 xe XG_TD_BG_DEPTH=4
run td_onlybg xe XG_TD_ONLY_BG=1
run td_nobg xe XG_TD_ONLY_BG=2
