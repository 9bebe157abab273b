#!/bin/bash
# knobs on the hidden-1024 bf16 configuration (diag library)
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$GRAFT_REPO_ROOT/controllable_xgating_amd/lib/libxgate_hip_diag.so
run() { local name=$1; shift
  local out=$(env "$@" timeout 200 python bench.py --no-secondary --no-cpu-baseline --no-pmc --workload xe5 --precision bf16 --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['in_situ_us_per_step'])")
  echo "$name : $out"; }
run base A=1
run nw8 XG_SK_NW=8
run formD XG_XE_FORM=D
run formF XG_XE_FORM=F
run c1ks4 XG_C1_KS=4
run c1ks8 XG_C1_KS=8
run c1lag1 XG_C1_LAG=1
run c1lag21 XG_C1_LAG=21
run nobg XG_GEMM_NO_BG=1
run fwdbg0 XG_FWD_BG=0
run bwdth6 XG_BWD_TH=6
run bwdth14 XG_BWD_TH=14
run base A=1
