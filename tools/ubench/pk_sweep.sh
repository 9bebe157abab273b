export XG_LIBRARY=/root/repo/controllable_xgating_amd/lib/libxgate_hip_diag.so
export XG_GEMM_SHAPES="wgrad,embed,PRE,vproj,dX,rollout logits NT 128"
for t in 0 1 2; do
  echo "== XG_PK_TILE=$t"
  XG_PK_TILE=$t python tools/ubench/gemm_bench.py one 0 | python -c "
import sys, json
for line in sys.stdin:
    i = line.index('{'); d = json.loads(line[i:])
    for k, v in d.items(): print('  %-40s %8.1f us %7.1f TF  err %.1e' % (k, v[0], v[1], v[2]))
"
done
