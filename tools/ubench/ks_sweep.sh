#!/bin/bash
# cross-workgroup split caps of the reverse-time loop's launches A (daf || dh2 partial) and B (dh2 += dp W + cell-2 backward epilogue)
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$GRAFT_REPO_ROOT/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { local name=$1; shift
  local out=$(env "$@" timeout 200 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "$name : $out"; }
run base A=1
for k in 1 2 4; do run B_ks$k XG_B_KS=$k; done
for k in 1 2; do run A_ks$k XG_A_KS=$k; done
run B2_A2 XG_B_KS=2 XG_A_KS=2
run nosplit XG_NO_SPLITK=1
run base A=1
