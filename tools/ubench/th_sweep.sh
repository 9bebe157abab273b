#!/bin/bash
# where the dH = dlogits W product is split between the exposed late part (main stream) and the background early part (XG_BWD_TH),
# and the forward counterpart for the logits (XG_FWD_TH).  diag library.
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$GRAFT_REPO_ROOT/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { local name=$1; shift
  local out=$(env "$@" timeout 200 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('in_situ_us_per_step'))")
  echo "$name : $out"; }
run base A=1
for th in 4 6 8 9 11 12 14; do run bwd_th$th XG_BWD_TH=$th; done
for th in 6 8 12 14 16; do run fwd_th$th XG_FWD_TH=$th; done
run base A=1
