#!/bin/bash
# in-situ durations of the weight-gradient kernels with / without gemm_td_kernel on every eligible product (diag library)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export XG_LIBRARY=$GRAFT_REPO_ROOT/controllable_xgating_amd/lib/libxgate_hip_diag.so
for cfg in "base A=1" "td_all XG_TD_ALL=1"; do
  set -- $cfg; name=$1; shift
  rm -rf gpurun_out/tdp
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tdp -- python bench.py --no-cpu-baseline --no-pmc --no-secondary --steps 10 --warmup 3 > /dev/null 2>&1
  echo "== $name"
  python tools/prof_summary.py gpurun_out/tdp /tmp/s.txt 23 > /dev/null; head -3 /tmp/s.txt | tail -2; grep -E "gemm_td|gemm_kernel<64, 64, false, false|gemm_w1_kernel<64, 64, 128, false, false|gemm_pk_kernel<128, 128, false, false" /tmp/s.txt | head -8
done
rm -rf gpurun_out/tdp
