#!/bin/bash
# instruction / wait counters of the decoder step's launches (stand-alone step group, B = 128) for one library build.
# usage: step_pmc.sh <out.txt> [lib.so relative to controllable_xgating_amd/lib]
cd ${GRAFT_REPO_ROOT:-.}
OUT=$(mktemp -d /tmp/steppmc.XXXX)
[ -n "$2" ] && export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/$2
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc1 -- python $R/tools/step_group_run.py 40 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/pmc2 -- python $R/tools/step_group_run.py 40 > $OUT/pmc2.log 2>&1
cd $R
python tools/pmc_summary.py $1 "decoder-step launch group (tools/step_group_run.py, B=128), library ${2:-libxgate_hip.so}: rocprofv3 --pmc, two passes" $OUT/pmc1 $OUT/pmc2 | grep -A16 "skf_kernel"
rm -rf $OUT
