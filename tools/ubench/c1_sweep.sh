#!/bin/bash
# chain-1 knobs of the decoder backward (diag library): cross-workgroup split cap, wave priority, event lag.
# usage: c1_sweep.sh  -> one line per setting: ms_per_step
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$GRAFT_REPO_ROOT/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { # name, env...
  local name=$1; shift
  local out=$(env "$@" timeout 200 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('in_situ_us_per_step'), d.get('parity_loss_delta'))")
  echo "$name : $out"
}
run base A=1
run base2 A=1
run ks4 XG_C1_KS=4
run ks2 XG_C1_KS=2
run ks1 XG_C1_KS=1
run lowprio XG_C1_LOWPRIO=1
run ks2_lowprio XG_C1_KS=2 XG_C1_LOWPRIO=1
run lag2 XG_C1_LAG=2
run lag3 XG_C1_LAG=3
run lag3_ks2 XG_C1_LAG=3 XG_C1_KS=2
run lag3_ks2_lp XG_C1_LAG=3 XG_C1_KS=2 XG_C1_LOWPRIO=1
run lag5_ks2 XG_C1_LAG=5 XG_C1_KS=2
