for i in 1 2; do
for lib in libxgate_hip_diag.so libxgate_hip_fragpipe.so; do
  XG_LIBRARY=/root/repo/controllable_xgating_amd/lib/$lib python bench.py --no-pmc --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'])"
done; done
XG_GEMM_SHAPES="logits fwd,dW_logit,dH NN K=20000" XG_GEMM_FORCE_BG=1 XG_LIBRARY=/root/repo/controllable_xgating_amd/lib/libxgate_hip_diag.so python tools/ubench/gemm_bench.py one 0 2>/dev/null | tail -1
XG_GEMM_SHAPES="logits fwd,dW_logit,dH NN K=20000" XG_GEMM_FORCE_BG=1 XG_LIBRARY=/root/repo/controllable_xgating_amd/lib/libxgate_hip_fragpipe.so python tools/ubench/gemm_bench.py one 0 2>/dev/null | tail -1
