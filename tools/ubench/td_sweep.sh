#!/bin/bash
# gemm_td_kernel knobs (diag library) on the weight-gradient shapes
cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$GRAFT_REPO_ROOT/controllable_xgating_amd/lib/libxgate_hip_diag.so
export XG_GEMM_SHAPES="TN"
run() { local name=$1; shift; echo "$name : $(env "$@" python tools/ubench/gemm_bench.py one 0 2>/dev/null | tail -1 | cut -c1-200)"; }
run base A=1
run no_td XG_GEMM_NO_TD=1
run depth12 XG_TD_DEPTH=12
run ks2 XG_TD_KS=2
run ks2_d12 XG_TD_KS=2 XG_TD_DEPTH=12
run ks4 XG_TD_KS=4
