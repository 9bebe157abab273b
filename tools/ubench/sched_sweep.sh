#!/bin/bash
# scheduling switches of the XE iteration (diag library).  usage: sched_sweep.sh -> one line per setting
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$GRAFT_REPO_ROOT/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { # name, env...
  local name=$1; shift
  local out=$(env "$@" timeout 200 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('in_situ_us_per_step'), d.get('final_loss'))")
  echo "$name : $out"
}
run base A=1
run base2 A=1
run defer XG_DEFER_WG=1
run toklate XG_TOK_LATE=1
run both XG_DEFER_WG=1 XG_TOK_LATE=1
run both_c1 XG_DEFER_WG=1 XG_TOK_LATE=1 XG_C1_LAG=5 XG_C1_KS=2
run c1 XG_C1_LAG=5 XG_C1_KS=2
run c1b XG_C1_LAG=7 XG_C1_KS=2 XG_C1_LOWPRIO=1
run c1c XG_C1_LAG=21 XG_C1_KS=2
run c1d XG_C1_LAG=4 XG_C1_KS=4
