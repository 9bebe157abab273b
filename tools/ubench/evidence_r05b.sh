#!/bin/bash
# evidence of the round's last session in one GPU call: gpurun_out/r05b/*
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05b; mkdir -p $OUT
python bench.py > $OUT/r05_bench_line.json 2> $OUT/bench_line.err
python bench.py --no-cpu-baseline --workload xe5 --precision bf16 > $OUT/r05_bench_line_xe5_bf16.json 2>/dev/null
python bench.py --no-cpu-baseline --precision bf16x3 > $OUT/r05_bench_line_bf16x3.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/xe5 -- python bench.py --no-cpu-baseline --no-pmc --no-secondary --workload xe5 --precision bf16 --steps 8 --warmup 2 > $OUT/xe5.log 2>&1
python tools/prof_summary.py $OUT/xe5 $OUT/r05_xe5_bf16_kernel_stats.txt 15 > /dev/null
python tools/timeline.py $OUT/xe5 > $OUT/r05_xe5_bf16_timeline.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/x3 -- python bench.py --no-cpu-baseline --no-pmc --no-secondary --precision bf16x3 --steps 8 --warmup 2 > $OUT/x3.log 2>&1
python tools/prof_summary.py $OUT/x3 $OUT/r05_xe_bf16x3_kernel_stats.txt 23 > /dev/null
python tools/timeline.py $OUT/x3 > $OUT/r05_xe_bf16x3_timeline.txt 2>&1
rm -rf $OUT/xe5 $OUT/x3
python tools/ubench/gemm_bench.py > $OUT/r05_gemm_bench_raw.txt 2>/dev/null
bash tools/ubench/g16_ab.sh $OUT > /dev/null 2>&1
bash tools/ubench/g16_burst.sh $OUT > /dev/null 2>&1
bash tools/ubench/x3_planes.sh $OUT > /dev/null 2>&1
ls -la $OUT
