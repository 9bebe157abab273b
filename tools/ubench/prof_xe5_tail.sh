cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_xe5; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/xe5 -- python bench.py --no-cpu-baseline --no-pmc --no-secondary --workload xe5 --precision bf16 --steps 8 --warmup 2 > $OUT/xe5.log 2>&1
python tools/timeline.py $OUT/xe5 2 > $OUT/tl.txt 2>&1
python tools/timeline.py $OUT/xe5 2 ${DETAIL:-4600 7400} 2>&1 | sed -n '/^detail/,$p' > $OUT/xe5_detail.txt
rm -rf $OUT/xe5
sed -n 4,9p $OUT/tl.txt; grep -n "skf_kernel<4, 1, false, 1> x3[0-9]\|x40" $OUT/tl.txt | cut -c1-160
cat $OUT/xe5_detail.txt | grep -v "skf_kernel\|lstm_bwd2" | cut -c1-120 | head -100
