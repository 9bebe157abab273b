#!/bin/bash
# kernel stats of the SCST iteration (rocprofv3 --kernel-trace --stats) -> gpurun_out/r6/<name>_scst_kernel_stats.txt
name=${1:-scst}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -- python bench.py --no-cpu-baseline --no-pmc --workload scst --steps 10 --warmup 3 > $OUT/$name.log 2>&1
python tools/prof_summary.py $OUT/prof_$name $OUT/${name}_scst_kernel_stats.txt 18 > /dev/null
rm -rf $OUT/prof_$name
head -30 $OUT/${name}_scst_kernel_stats.txt | cut -c1-200
