#!/bin/bash
# A/B of the headline XE line (round 6): product library twice, then the diag library with each given switch.  usage: xe_ab.sh [ENVVAR=1 ...]
cd "$(dirname "$0")/../.."
out=gpurun_out/r6_xe_ab.txt
: > $out
run() { label=$1; shift
  env "$@" python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read())
print('$label', 'ms', o['ms_per_step'], 'step us', o['roofline']['avg_launch_us'], 'in-situ', o['roofline']['in_situ_us_per_step'])" | tee -a $out
}
run product; run product
for v in "$@"; do run "diag $v" XG_LIBRARY=controllable_xgating_amd/lib/libxgate_hip_diag.so $v; done
