#!/bin/bash
# kernel trace of the training iteration + timeline (whole) + detail window.  usage: prof_iter.sh [from_us to_us] (extra bench args via BENCH_ARGS)
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_iter
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python bench.py --no-cpu-baseline --no-pmc --steps 10 --warmup 3 $BENCH_ARGS > $OUT/bench.log 2>&1 < /dev/null
python tools/timeline.py $OUT/bench > $OUT/timeline.txt 2>&1
if [ -n "$1" ]; then python tools/timeline.py $OUT/bench 1 $1 $2 > $OUT/detail.txt 2>&1; fi
python tools/prof_summary.py $OUT/bench $OUT/kernel_stats.txt 18 > /dev/null 2>&1
rm -rf $OUT/bench
tail -1 $OUT/bench.log | cut -c1-200
