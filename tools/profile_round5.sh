#!/bin/bash
# Round-5 evidence in one GPU session (reduced form of tools/profile_round.sh): gpurun_out/prof_r05/* -> copy into profiles/.
set -u
R=r05
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# 1. full iteration (configs[1]): kernel trace + stats + timeline
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python bench.py --no-cpu-baseline --no-pmc --no-secondary --steps 10 --warmup 3 > $OUT/bench.log 2>&1
python tools/prof_summary.py $OUT/bench $OUT/${R}_bench_kernel_stats.txt 23 > /dev/null
python tools/timeline.py $OUT/bench > $OUT/${R}_bench_timeline.txt 2>&1
python tools/timeline.py $OUT/bench 1 2700 3000 2>&1 | sed -n '/^detail/,$p' >> $OUT/${R}_bench_timeline.txt
# 2. decoder-step launch group: kernel stats, per-launch durations, traffic, counters
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/step -- python tools/step_group_run.py 200 > $OUT/step.log 2>&1
python tools/prof_summary.py $OUT/step $OUT/${R}_step_group_kernel_stats.txt 210 > /dev/null
python tools/step_trace.py $OUT/step >> $OUT/${R}_step_group_kernel_stats.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmcF -- python tools/step_group_run.py 40 > $OUT/pmcF.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmcW -- python tools/step_group_run.py 40 > $OUT/pmcW.log 2>&1
python tools/pmc_traffic.py $OUT/pmcF $OUT/pmcW $OUT/${R}_step_traffic.json > /dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc1 -- python tools/step_group_run.py 40 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/pmc2 -- python tools/step_group_run.py 40 > $OUT/pmc2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc3 -- python tools/step_group_run.py 40 > $OUT/pmc3.log 2>&1
python tools/pmc_summary.py $OUT/${R}_step_pmc.txt "decoder-step launch group (tools/step_group_run.py, B=128): rocprofv3 --pmc, three passes" $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 > /dev/null
# 3. SCST iteration (configs[2]) and the bf16 configuration (configs[4] shape)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/scst -- python bench.py --no-cpu-baseline --no-pmc --workload scst --steps 10 --warmup 3 > $OUT/scst.log 2>&1
python tools/prof_summary.py $OUT/scst $OUT/${R}_scst_kernel_stats.txt 18 > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/xe5 -- python bench.py --no-cpu-baseline --no-pmc --workload xe5 --precision bf16 --steps 8 --warmup 2 > $OUT/xe5.log 2>&1
python tools/prof_summary.py $OUT/xe5 $OUT/${R}_xe5_bf16_kernel_stats.txt 15 > /dev/null
# 4. large products alone
python tools/ubench/gemm_bench.py > $OUT/${R}_gemm_bench_raw.txt 2>&1
# 5. in-kernel stamps of the step's launches and instruction counts by region (variant libraries built beforehand: tools/ubench/build_ablate.py,
#    __graft_entry__.build_variant('sktrace', ['-DSK_TRACE']); skipped when they are not there)
if [ -f controllable_xgating_amd/lib/libxgate_hip_sktrace.so ] && [ -f controllable_xgating_amd/lib/libxgate_hip_abl1.so ]; then
{ echo "# tools/sk_trace_all.sh on the final round-5 kernel: in-kernel stamps of the decoder step's launches, B = 128 (same columns as r05_sk_trace_before.txt)"; SK_PRECS="fp32 bf16" bash tools/sk_trace_all.sh 2>&1 | grep -v amdgpu.ids; echo; echo "# launches inside the training iteration (tools/sk_trace_iter.py): 2 jobs x 256 x 256 threads = the encoder's backward recurrence (last launch), 1 x 128 x 256 = chain 1"; XG_LIBRARY=$GRAFT_REPO_ROOT/controllable_xgating_amd/lib/libxgate_hip_sktrace.so python tools/sk_trace_iter.py 2>&1 | grep -v amdgpu.ids; } > $OUT/${R}_sk_trace_after.txt
{ echo "# tools/ubench/ablate_step.sh: rocprofv3 --pmc instruction counts of the step's launches for builds that return behind (6) the descriptor round,"; echo "# (1) tile decode, (2) epilogue-operand requests, (5) first segment set-up + first operand request, (3) the K loops, (4) the reduction; per-dispatch totals"; echo "# (skf_kernel<8,0,true,1> = cell 2's launch: 4096 waves of which 2048 belong to product tiles)"; bash tools/ubench/ablate_step.sh 2>&1; } > $OUT/${R}_step_ablation.txt
fi
# 6. the bench lines themselves (un-profiled)
python bench.py > $OUT/${R}_bench_line.json 2> $OUT/bench_line.err
python bench.py --no-cpu-baseline --workload scst > $OUT/${R}_bench_line_scst.json 2>/dev/null
python bench.py --no-cpu-baseline --workload xe5 --precision bf16 > $OUT/${R}_bench_line_xe5_bf16.json 2>/dev/null
python bench.py --no-cpu-baseline --precision bf16x3 > $OUT/${R}_bench_line_bf16x3.json 2>/dev/null
XG_FORCE_DIST=2 python bench.py --no-cpu-baseline --no-pmc --no-secondary > $OUT/${R}_bench_line_one_rank_rccl.json 2>/dev/null
rm -rf $OUT/bench $OUT/step $OUT/scst $OUT/xe5 $OUT/pmcF $OUT/pmcW $OUT/pmc1 $OUT/pmc2 $OUT/pmc3
ls -la $OUT
