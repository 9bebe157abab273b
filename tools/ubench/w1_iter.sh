for i in 1 2; do
for v in "" "XG_GEMM_NO_W1=1"; do
  echo "== $v"
  env $v XG_LIBRARY=/root/repo/controllable_xgating_amd/lib/libxgate_hip_diag.so python bench.py --no-pmc --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done; done
