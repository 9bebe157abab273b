#!/bin/bash
# A/B of the SCST line (round 6): product library, diag library with / without a switch.  usage: scst_ab.sh [ENVVAR=1 ...]
cd "$(dirname "$0")/../.."
out=gpurun_out/r6_scst_ab.txt
: > $out
run() { # label env...
  label=$1; shift
  env "$@" python bench.py --workload scst --no-pmc --steps 10 --warmup 5 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read())
print('$label', 'ms', o['ms_per_step'], 'in-situ us/step', o['roofline']['in_situ_us_per_step'], 'parity', o.get('parity_loss_delta'))" | tee -a $out
}
run product
run product
for v in "$@"; do
  run "diag $v" XG_LIBRARY=controllable_xgating_amd/lib/libxgate_hip_diag.so $v
done
