#!/bin/bash
# Soak of the round's tree: thousands of iterations on one fixed synthetic batch per line (bench.py --steps N); what it is for: the
# split launches' hand-off spins on tagged granules and the rollout's token choice runs inside the step -- no hang, no
# drift of the iteration time, the loss keeps falling.  Writes gpurun_out/soak.txt.
cd "$(dirname "$0")/../.."
out=gpurun_out/soak.txt; : > $out
run() { label=$1; shift
  timeout 900 python bench.py --no-cpu-baseline --no-pmc --no-secondary "$@" 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s steps %5d  ms/iteration %.3f  %s' % ('$label', o['steps'], o['ms_per_step'], {k: o[k] for k in ('parity_loss_delta', 'final_loss', 'loss_first', 'loss_last') if k in o}))" | tee -a $out
}
run "XE fp32" --steps 3000 --warmup 5
run "XE fp32 drop 0.5" --steps 1500 --warmup 5 --drop 0.5
run "XE split-bf16" --steps 1500 --warmup 5 --precision bf16x3
run "SCST" --workload scst --steps 1000 --warmup 5
run "configs[4] bf16" --workload xe5 --precision bf16 --steps 1000 --warmup 5
