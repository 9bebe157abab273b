#!/bin/bash
# A/B of two builds of the library on one box: usage ab_lib.sh <libA.so> <libB.so> [workload] [reps]  (paths relative to controllable_xgating_amd/lib)
cd $GRAFT_REPO_ROOT
A=$1; B=$2; WL=${3:-xe}; N=${4:-3}
run() { local lib=$1
  XG_LIBRARY=$GRAFT_REPO_ROOT/controllable_xgating_amd/lib/$lib timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --workload $WL --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['roofline'].get('in_situ_us_per_step'))"; }
for i in $(seq $N); do run $A; run $B; done
