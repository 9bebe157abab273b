#!/bin/bash
# What do N extra scalar instructions in every wave's prologue cost a skinny launch?  (the price of the interpreter's
# ~226 scalar instructions in front of the first operand request, measured by ADDING to them.)  Builds: -DSKF_PAD_SALU=128 / 256.
L=controllable_xgating_amd/lib
for rep in 1 2; do
for v in product pad128 pad256; do
    lib=""; [ $v != product ] && lib="XG_LIBRARY=$L/libxgate_hip_$v.so"
    env $lib python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('%-8s XE %.3f ms | step group in situ %.2f us, by arithmetic %s | SCST %.3f ms, rollout step %s us' % ('$v', d['ms_per_step'], r['avg_launch_us'],
      r.get('step_us_by_arithmetic'), d['secondary']['scst']['ms_per_step'], d['secondary']['scst'].get('roofline', {}).get('avg_launch_us')))"
done; done
