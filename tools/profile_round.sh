#!/bin/bash
# One GPU session that regenerates the round's rocprofv3 evidence under gpurun_out/prof_$1 (copy the summaries into
# profiles/ afterwards).  usage: tools/profile_round.sh r02
set -u
R=${1:-rXX}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# 1. full iteration (configs[1]): kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python bench.py --no-cpu-baseline --no-pmc --no-secondary --steps 10 --warmup 3 > $OUT/bench.log 2>&1
python tools/prof_summary.py $OUT/bench $OUT/${R}_bench_kernel_stats.txt 23 > /dev/null
python tools/timeline.py $OUT/bench > $OUT/${R}_bench_timeline.txt 2>&1
python tools/timeline.py $OUT/bench 1 2950 3300 2>&1 | sed -n '/^detail/,$p' >> $OUT/${R}_bench_timeline.txt
# 2. decoder-step launch group: kernel stats, per-launch durations, PMC passes
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
# 4. large products; bf16 operands in memory vs converted on the fly; L2 / fabric fill rates
python tools/ubench/gemm_bench.py > $OUT/${R}_gemm_bench_raw.txt 2>&1
python tools/ubench/gemm16_bench.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" > $OUT/${R}_gemm_bf16_operands.txt
(cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fill_bench fill_bench.hip 2>/dev/null; timeout 200 /tmp/fill_bench) > $OUT/${R}_fill_bench.txt 2>&1
# 4a. fp32 GEMM yardsticks: vendor library on the same shapes, MFMA issue rate, in-kernel stamps of the one-workgroup-per-CU kernel
{ echo "# tools/ubench/lib_gemm_bench.py: the vendor fp32 GEMM (torch.mm -> hipBLASLt / Tensile) on the same shapes, as a yardstick (not used by the product)."; python tools/ubench/lib_gemm_bench.py 2>&1 | grep -v "amdgpu.ids"; } > $OUT/${R}_vendor_gemm_yardstick.txt
(cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate mfma_rate.hip 2>/dev/null; { echo "# tools/ubench/mfma_rate.hip: issue rate of the fp32 MFMAs with nothing else in the loop (256 workgroups)."; timeout 100 /tmp/mfma_rate; }) > $OUT/${R}_mfma_rate.txt 2>&1
(/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DXG_DIAG -DW1_TRACE -Iinclude tools/ubench/w1_ubench.hip -o /tmp/w1_ubench 2>/dev/null; { echo "# tools/ubench/w1_ubench.hip (-DW1_TRACE): in-kernel stamps of gemm_w1_kernel -- prologue, slab loop (shader clocks), epilogue -- per workgroup"; timeout 100 /tmp/w1_ubench; }) > $OUT/${R}_gemm_w1_trace.txt 2>&1
python tools/select_check.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" > $OUT/${R}_rollout_select_check.txt
# 4b. the round's experiments: the step as one dataflow launch (in-kernel timeline), HIP-graph capture variants
python tools/dstep_trace.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" > $OUT/${R}_dstep_timeline.txt
for b in 8 32 64 128 256; do echo "rows $b: $(DS_B=$b timeout 300 python tools/dstep_check.py 2>&1 | grep 'us per step')"; done > $OUT/${R}_dstep_vs_three_launches.txt
python tools/graph_probe.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" > $OUT/${R}_graph_probe.txt
# 4c. round 4: CU-masked streams probe, chain-1 / scheduling / split-point sweeps, arithmetic-mode reproducibility of the step
(cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/cumask_probe cumask_probe.hip 2>/dev/null; { echo "# tools/ubench/cumask_probe.hip: hipExtStreamCreateWithCUMask -- which mask bit is which (XCD, SE, CU); a latency-bound chain beside an MFMA background kernel, shared CUs vs complementary masks"; timeout 120 /tmp/cumask_probe; }) > $OUT/${R}_cumask_probe.txt 2>&1
{ echo "# tools/ubench/c1_sweep.sh (diag library): chain 1 of the decoder backward -- split cap / wave priority / steps per event: ms per iteration, in-situ us per forward step"; bash tools/ubench/c1_sweep.sh 2>&1 | grep " : "; } > $OUT/${R}_c1_sweep.txt
{ echo "# tools/ubench/sched_sweep.sh (diag library): deferred decoder weight gradients (XG_DEFER_WG), late token side (XG_TOK_LATE), chain-1 knobs"; bash tools/ubench/sched_sweep.sh 2>&1 | grep " : "; } > $OUT/${R}_sched_sweep.txt
{ echo "# tools/ubench/th_sweep.sh (diag library): split point of dH = dlogits W (XG_BWD_TH) and of the forward logits (XG_FWD_TH)"; bash tools/ubench/th_sweep.sh 2>&1 | grep " : "; } > $OUT/${R}_th_sweep.txt
{ echo "# tools/step_mode_check.py: xg_step_fwd x 1/2/3 in place, gemm_mode 3 (split-bf16 step products) against gemm_mode 0; 20 repetitions each"; python tools/step_mode_check.py 2>&1 | grep "^steps"; } > $OUT/${R}_step_mode_check.txt
{ echo "# XG_XE_FORM=F (cell 1 one step ahead) against the default teacher-forced step form (diag library): ms per iteration, in-situ us per step"; for f in D F D F; do XG_LIBRARY=$GRAFT_REPO_ROOT/controllable_xgating_amd/lib/libxgate_hip_diag.so XG_XE_FORM=$f python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('form $f :', d['ms_per_step'], d['roofline'].get('in_situ_us_per_step'))"; done; } > $OUT/${R}_xe_form_f.txt
# 5. the bench lines themselves (un-profiled)
python bench.py > $OUT/${R}_bench_line.json 2> $OUT/bench_line.err
python bench.py --no-cpu-baseline --workload scst > $OUT/${R}_bench_line_scst.json 2>/dev/null
python bench.py --no-cpu-baseline --workload xe5 --precision bf16 > $OUT/${R}_bench_line_xe5_bf16.json 2>/dev/null
python bench.py --no-cpu-baseline --precision bf16x3 > $OUT/${R}_bench_line_bf16x3.json 2>/dev/null
python bench.py --no-cpu-baseline --no-pmc --no-secondary --graph > $OUT/${R}_bench_line_graph.json 2>/dev/null
XG_FORCE_DIST=2 python bench.py --no-cpu-baseline --no-pmc --no-secondary > $OUT/${R}_bench_line_one_rank_rccl.json 2>/dev/null
# keep the merge small: raw traces stay on the box
rm -rf $OUT/bench $OUT/step $OUT/scst $OUT/xe5 $OUT/pmcF $OUT/pmcW $OUT/pmc1 $OUT/pmc2 $OUT/pmc3
ls -la $OUT
