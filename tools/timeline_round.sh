#!/bin/bash
# Timeline of one training iteration (configs[1]) -> gpurun_out/prof_$R/${R}_bench_timeline.txt   usage: R=r05 bash tools/timeline_round.sh
set -u
R=${R:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python bench.py --no-cpu-baseline --no-pmc --no-secondary --steps 10 --warmup 3 > $OUT/bench.log 2>&1
python tools/timeline.py $OUT/bench > $OUT/${R}_bench_timeline.txt 2>&1
python tools/timeline.py $OUT/bench 1 2700 3000 2>&1 | sed -n '/^detail/,$p' >> $OUT/${R}_bench_timeline.txt
rm -rf $OUT/bench
