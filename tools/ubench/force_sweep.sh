# fp32 tile / split-K sweep of xg_gemm.hip's one-tile-per-workgroup kernels on the mid-size shapes (XG_GEMM_FORCE, diag library)
export XG_LIBRARY=${XG_LIBRARY:-/root/repo/controllable_xgating_amd/lib/libxgate_hip_diag.so}
export XG_GEMM_SHAPES="${XG_GEMM_SHAPES:-wgrad,embed,PRE,vproj,dX}"
for f in ${FORCES:-auto 64,1 64,2 64,3 128,1 128,2 128,4}; do
  if [ "$f" = auto ]; then unset XG_GEMM_FORCE; else export XG_GEMM_FORCE=$f; fi
  python tools/ubench/gemm_bench.py one 0 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if '{' not in line: continue
    i = line.index('{'); d = json.loads(line[i:])
    print('%-8s' % '$f', '  '.join('%s %.1f us %.0f TF' % (k.split()[0] + ' ' + k.split()[1], v[0], v[1]) for k, v in d.items()))
"
done
