# iteration time under a few switches of the -DXG_DIAG library (re-checked on the final tree of round 3: the defaults stand)
for v in "" "XG_WG_CHUNKS=1" "XG_WG_CHUNKS=3" "XG_GEMM_NO_W1=1" "XG_FWD_BG=1" "XG_PK_G=128" "XG_PK_G=192" "XG_PK_G=384" ""; do
  env $v XG_LIBRARY=/root/repo/controllable_xgating_amd/lib/libxgate_hip_diag.so python bench.py --no-pmc --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v'.ljust(22), d['ms_per_step'])"
done
