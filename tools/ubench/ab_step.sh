#!/bin/bash
# A/B of two library builds on the stand-alone step group and the iteration.  usage: ab_step.sh libA.so libB.so [reps]
cd $GRAFT_REPO_ROOT
for rep in 1 2 ${3:+3}; do for lib in $1 $2; do
  XG_LIBRARY=$GRAFT_REPO_ROOT/controllable_xgating_amd/lib/$lib timeout 200 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', d['ms_per_step'], 'step', r['avg_launch_us'], 'in situ', r['in_situ_us_per_step'])"
done; done
