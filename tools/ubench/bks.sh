#!/bin/bash
# reverse-time loop per step (rocprofv3 timeline) under split caps of its launches A / B (diag library)
cd $GRAFT_REPO_ROOT
for cfg in "XG_B_KS=0 XG_A_KS=0" "XG_B_KS=1" "XG_B_KS=2" "XG_A_KS=2" "XG_A_KS=1 XG_B_KS=1"; do
  OUT=$(mktemp -d /tmp/bks.XXXX)
  ( cd /tmp; export TMPDIR=/tmp; env $cfg XG_LIBRARY=$GRAFT_REPO_ROOT/controllable_xgating_amd/lib/${BKS_LIB:-libxgate_hip_diag.so} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pmc --no-secondary --steps 10 --warmup 3 > $OUT/bench.log 2>&1 < /dev/null )
  echo "== $cfg : $(tail -1 $OUT/bench.log | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)"
  python tools/timeline.py $OUT/bench | grep "reverse-time"
  rm -rf $OUT
done
