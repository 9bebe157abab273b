export XG_LIBRARY=$PWD/controllable_xgating_amd/lib/libxgate_hip_diag.so
export XG_GEMM_SHAPES="logits,dH NN,enc dW_hh"
for c in 642 844; do for d in 0 5 6; do
  echo "== XG_G16_CFG=$c XG_G16_DBG=$d"
  XG_G16_CFG=$c XG_G16_DBG=$d python tools/ubench/gemm16_bench.py both 2>/dev/null | cut -d'|' -f1 | sed 's/err.*//'
done; done
