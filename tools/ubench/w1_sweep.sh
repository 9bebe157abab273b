# the one-workgroup-per-CU kernel (xg_gemm.hip: gemm_w1_kernel) against the older routes on the mid-size shapes; XG_W1_TILE forces a tile
export XG_LIBRARY=${XG_LIBRARY:-/root/repo/controllable_xgating_amd/lib/libxgate_hip_diag.so}
export XG_GEMM_SHAPES="${XG_GEMM_SHAPES:-wgrad,embed,PRE,vproj,dX}"
run() {
  python tools/ubench/gemm_bench.py one 0 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if '{' not in line: continue
    i = line.index('{'); d = json.loads(line[i:])
    print('%-10s' % '$1', '  '.join('%s %.1f us %.0f TF %.0e' % (k.split()[0] + ' ' + k.split()[1], v[0], v[1], v[2]) for k, v in d.items()))
"
}
XG_GEMM_NO_W1=1 run old
run auto
for t in ${TILES:-64,64 64,128 128,64 128,128 128,192 192,128 128,256 256,128}; do XG_W1_TILE=$t run $t; done
