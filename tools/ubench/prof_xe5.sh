cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_xe5; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/xe5 -- python bench.py --no-cpu-baseline --no-pmc --no-secondary --workload xe5 --precision bf16 --steps 8 --warmup 2 > $OUT/xe5.log 2>&1
python tools/prof_summary.py $OUT/xe5 $OUT/r05_xe5_bf16_kernel_stats.txt 15 > /dev/null
python tools/timeline.py $OUT/xe5 > $OUT/r05_xe5_bf16_timeline.txt 2>&1
python tools/timeline.py $OUT/xe5 1 ${DETAIL:-5900 7200} 2>&1 | sed -n '/^detail/,$p' > $OUT/xe5_detail.txt
rm -rf $OUT/xe5
cat $OUT/r05_xe5_bf16_timeline.txt | cut -c1-230 | sed -n 4,50p
