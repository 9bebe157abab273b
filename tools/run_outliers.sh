OUT=$GRAFT_REPO_ROOT/gpurun_out/outl
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/b -- python bench.py --no-cpu-baseline --no-pmc --steps 10 --warmup 3 $BENCH_ARGS > $OUT/bench.log 2>&1 < /dev/null
python tools/outliers.py $OUT/b 2.0 > $OUT/outliers.txt 2>&1
rm -rf $OUT/b
cat $OUT/outliers.txt | head -60
