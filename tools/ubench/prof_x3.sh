cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_x3; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/x3 -- python bench.py --no-cpu-baseline --no-pmc --no-secondary --precision bf16x3 --steps 8 --warmup 2 > $OUT/x3.log 2>&1
python tools/prof_summary.py $OUT/x3 $OUT/r05_xe_bf16x3_kernel_stats.txt 23 > /dev/null
python tools/timeline.py $OUT/x3 > $OUT/r05_xe_bf16x3_timeline.txt 2>&1
rm -rf $OUT/x3
head -36 $OUT/r05_xe_bf16x3_kernel_stats.txt | cut -c1-150
cat $OUT/r05_xe_bf16x3_timeline.txt | head -60
