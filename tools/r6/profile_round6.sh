#!/bin/bash
# Round-6 evidence in one GPU session: gpurun_out/prof_r06/* -> copy into profiles/ (tools: prof_summary.py, timeline.py, pmc_*).
# Variant libraries are built beforehand in the build container:
#   python -c "import __graft_entry__ as g; g.build_variant('sktrace', ['-DSK_TRACE']); g.build_variant('vptrace', ['-DVP_TRACE']); g.build_variant('null', ['-DXG_NULL_LAUNCH'])"
set -u
R=r06
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
LIB=$GRAFT_REPO_ROOT/controllable_xgating_amd/lib
# 1. full iteration (configs[1]): kernel trace + stats + timeline
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python bench.py --no-cpu-baseline --no-pmc --no-secondary --steps 10 --warmup 3 > $OUT/bench.log 2>&1
python tools/prof_summary.py $OUT/bench $OUT/${R}_bench_kernel_stats.txt 23 > /dev/null
python tools/timeline.py $OUT/bench > $OUT/${R}_bench_timeline.txt 2>&1
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
# 3. SCST iteration (configs[2]), the bf16 configuration (configs[4] shape), the headline config at drop_prob_lm 0.5
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/scst -- python bench.py --no-cpu-baseline --no-pmc --workload scst --steps 10 --warmup 3 > $OUT/scst.log 2>&1
python tools/prof_summary.py $OUT/scst $OUT/${R}_scst_kernel_stats.txt 18 > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/xe5 -- python bench.py --no-cpu-baseline --no-pmc --workload xe5 --precision bf16 --steps 8 --warmup 2 > $OUT/xe5.log 2>&1
python tools/prof_summary.py $OUT/xe5 $OUT/${R}_xe5_bf16_kernel_stats.txt 15 > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/drop -- python bench.py --no-cpu-baseline --no-pmc --no-secondary --drop 0.5 --steps 10 --warmup 3 > $OUT/drop.log 2>&1
python tools/prof_summary.py $OUT/drop $OUT/${R}_drop05_kernel_stats.txt 18 > /dev/null
# 4. in-kernel stamps: the rollout step's first launch with the token choice as its prologue; the vocabulary product
if [ -f $LIB/libxgate_hip_sktrace.so ]; then XG_LIBRARY=$LIB/libxgate_hip_sktrace.so python tools/r6/sel_trace.py 2>&1 | grep -v amdgpu.ids | cut -c1-1500 > $OUT/${R}_rollout_l1_select_trace.txt; fi
if [ -f $LIB/libxgate_hip_vptrace.so ]; then XG_LIBRARY=$LIB/libxgate_hip_vptrace.so python tools/r6/vp_trace.py 2>&1 | grep -v amdgpu.ids > $OUT/${R}_vocab_part_trace.txt; fi
# 5. host side with eight launchers (product library and null-launch build), grid barriers
python tools/host8_enqueue.py 8 6 > $OUT/${R}_host8_enqueue.json 2> $OUT/host8.err
hipcc --offload-arch=gfx950 -O2 -o /tmp/grid_barrier tools/ubench/grid_barrier.hip > /dev/null 2>&1 && timeout 120 /tmp/grid_barrier 1000 > $OUT/${R}_grid_barrier.txt 2>&1
# 6. large products alone
python tools/ubench/gemm_bench.py > $OUT/${R}_gemm_bench.txt 2>&1
# 7. the bench lines themselves (un-profiled)
python bench.py > $OUT/${R}_bench_line.json 2> $OUT/bench_line.err
XG_FORCE_DIST=2 python bench.py --no-cpu-baseline --no-pmc --no-secondary > $OUT/${R}_bench_line_one_rank_rccl.json 2>/dev/null
rm -rf $OUT/bench $OUT/step $OUT/scst $OUT/xe5 $OUT/drop $OUT/pmcF $OUT/pmcW $OUT/pmc1 $OUT/pmc2 $OUT/pmc3
ls -la $OUT
