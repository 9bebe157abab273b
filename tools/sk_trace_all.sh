#!/bin/bash
# in-kernel phase timelines of the three launches of the decoder step (B = 128): L3 (cell 2: 2 jobs, 256 wide), L2 (3 jobs), L1 (3 jobs, 192 wide)
cd $GRAFT_REPO_ROOT
for cfg in "2 256" "3 256" "3 192"; do
  set -- $cfg
  export XG_EXTRA_FLAGS="-DSK_TRACE -DSK_TRACE_NJOBS=$1 -DSK_TRACE_GX=$2"
  echo "== njobs $1 gx $2"
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  SK_GX=$2 timeout 400 python tools/sk_trace_run.py 2>&1 | tail -6
done
