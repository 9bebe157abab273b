cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_xe5; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/xe5 -- python bench.py --no-cpu-baseline --no-pmc --no-secondary --workload xe5 --precision bf16 --steps 8 --warmup 2 > $OUT/xe5.log 2>&1
python tools/prof_summary.py $OUT/xe5 $OUT/r05_xe5_bf16_kernel_stats.txt 15 > /dev/null
rm -rf $OUT/xe5
head -40 $OUT/r05_xe5_bf16_kernel_stats.txt | cut -c1-150
