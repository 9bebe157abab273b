#!/bin/bash
# decoder weight gradients as BACKGROUND products (one persistent workgroup per CU) beside the reverse-time recurrences (diag library)
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$GRAFT_REPO_ROOT/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { # name, env...
  local name=$1; shift
  local out=$(env "$@" timeout 200 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('in_situ_us_per_step'), d.get('final_loss'))")
  echo "$name : $out"
}
run base A=1
run wgbg_min0 XG_WG_BG=1 XG_BG_MIN=0
run wgbg_min4 XG_WG_BG=1 XG_BG_MIN=4
run wgbg_min10 XG_WG_BG=1 XG_BG_MIN=10
run base2 A=1
run wgbg_min0_defer XG_WG_BG=1 XG_BG_MIN=0 XG_DEFER_WG=1
run wgbg_min0_chunks1 XG_WG_BG=1 XG_BG_MIN=0 XG_WG_CHUNKS=1
run wgbg_min0_chunks4 XG_WG_BG=1 XG_BG_MIN=0 XG_WG_CHUNKS=4
run chunks4 XG_WG_CHUNKS=4
