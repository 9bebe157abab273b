export XG_GEMM_SHAPES="logits fwd,dW_logit,dH NN K=20000,dH half NN 1408,PRE,vproj"
run() { python tools/ubench/gemm_bench.py one 0 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if '{' not in line: continue
    i = line.index('{'); d = json.loads(line[i:])
    print('%-10s' % '$1', '  '.join('%s %.1f us %.0f TF' % (' '.join(k.split()[:2]), v[0], v[1]) for k, v in d.items()))
"; }
XG_GEMM_NO_W1=1 XG_LIBRARY=/root/repo/controllable_xgating_amd/lib/libxgate_hip_diag.so run base
XG_GEMM_NO_W1=1 XG_LIBRARY=/root/repo/controllable_xgating_amd/lib/libxgate_hip_fragpipe.so run fragpipe
XG_GEMM_NO_W1=1 XG_LIBRARY=/root/repo/controllable_xgating_amd/lib/libxgate_hip_diag.so run base
XG_GEMM_NO_W1=1 XG_LIBRARY=/root/repo/controllable_xgating_amd/lib/libxgate_hip_fragpipe.so run fragpipe
