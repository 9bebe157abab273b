#!/bin/bash
# critical-path items (diag library): classifier head beside the late logits (always on), encoder gates side by side, dH split point
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$GRAFT_REPO_ROOT/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { local name=$1; shift
  local out=$(env "$@" timeout 200 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('in_situ_us_per_step'), d.get('final_loss'))")
  echo "$name : $out"; }
run base A=1
run no_enc_par XG_ENC_PAR=0
run enc_par_bwd XG_ENC_PAR_BWD=1
run base A=1
for th in 12 14 16 17 18; do run bwd_th$th XG_BWD_TH=$th; done
run base A=1
