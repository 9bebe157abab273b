#!/bin/bash
# dynamic instruction counts of the skinny kernel by region: builds with an early return behind (1) the tile decode, (2) the
# epilogue-operand requests, (3) the K loops, (4) the reduction (tools/ubench/build_ablate.py) against the full kernel.
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
for v in abl6 abl1 abl2 abl5 abl3 abl4 full; do
  lib=libxgate_hip_$v.so; [ $v = full ] && lib=libxgate_hip.so
  OUT=$(mktemp -d /tmp/abl.XXXX)
  ( cd /tmp; export TMPDIR=/tmp; XG_LIBRARY=$R/controllable_xgating_amd/lib/$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_INSTS_MFMA --output-format csv -d $OUT -- python $R/tools/step_group_run.py 30 > $OUT/log 2>&1 )
  echo "== $v"
  python tools/pmc_summary.py /dev/null "x" $OUT | grep -A8 "^skf_kernel"
  rm -rf $OUT
done
