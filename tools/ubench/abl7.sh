#!/bin/bash
# what the ticket / last-arriver phase of the split LSTM-backward tiles costs: the iteration with and without it (results wrong without)
cd $GRAFT_REPO_ROOT
for lib in libxgate_hip.so libxgate_hip_abl7.so; do
  OUT=$(mktemp -d /tmp/abl7.XXXX)
  ( cd /tmp; export TMPDIR=/tmp; XG_LIBRARY=$GRAFT_REPO_ROOT/controllable_xgating_amd/lib/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pmc --no-secondary --steps 10 --warmup 3 > $OUT/bench.log 2>&1 < /dev/null )
  echo "== $lib"; tail -1 $OUT/bench.log | cut -c1-120
  python tools/timeline.py $OUT/bench | grep "decoder loop"
  python tools/prof_summary.py $OUT/bench /dev/null 18 | grep "skf_kernel<4, 0, false>\|attn_bwd_split" | head -6
  rm -rf $OUT
done
