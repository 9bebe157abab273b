"""Diagnosis (round 6): B = 128, p = 0.5 -- is the HIP gradient of the pre-BatchNorm embedding matrices round-off-class?
Compares HIP (fp32) and the fp32 oracle against the oracle run in float64 on the same masks."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import paramgen as pg
from oracle import xgate_oracle as xo
from tests.util import CFG, make_model, to_dev

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
torch.set_num_threads(16)
d = pg.make_dims(**dict(CFG["c1"], B=B))
Pn = pg.make_params(d, logit_gain=8.0)
xn = pg.make_inputs(d, seed=int(sys.argv[3]) if len(sys.argv) > 3 else 0, ragged=True)
seed = 987654321

def run_oracle(dtype):
    torch.set_default_dtype(dtype)
    P = {k: torch.from_numpy(v).to(dtype).requires_grad_(True) for k, v in Pn.items()}
    xi = xo.to_torch_inputs(xn)
    xi = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in xi.items()}
    lo, _, _ = xo.forward_xe(P, xi["feats_rgb"], xi["feats_opfl"], xi["feat_mask"], xi["pos_feats"], xi["seq"], xi["seq_mask"],
                             train=True, p=p, seed=seed, running=None)
    loss = xo.lm_criterion(lo, xi["seq"], xi["seq_mask"])
    loss.backward()
    torch.set_default_dtype(torch.float32)
    return float(loss), {k: v.grad.double().numpy() if v.grad is not None else np.zeros(v.shape) for k, v in P.items()}

l64, g64 = run_oracle(torch.float64)
l32, g32 = run_oracle(torch.float32)
model = make_model(d, P=Pn, p_drop=p)
model.dropout_seed = seed
x = to_dev(xn)
loss = model.xe_loss(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], x["seq"], x["seq_mask"])
loss.backward()
torch.cuda.synchronize()
print("loss f64 %.8f  f32 oracle %.8f  hip %.8f" % (l64, l32, loss.item()))
print("%-52s %10s %12s %12s %12s" % ("param", "scale", "hip-f64", "o32-f64", "hip-o32"))
for name, prm in model.named_parameters():
    g = prm.grad.double().cpu().numpy()
    s = np.abs(g64[name]).max()
    if s < 1e-12: continue
    if len(sys.argv) > 4 and np.abs(g - g64[name]).max() / s < float(sys.argv[4]): continue
    print("%-52s %10.3e %12.3e %12.3e %12.3e" % (name, s, np.abs(g - g64[name]).max() / s, np.abs(g32[name] - g64[name]).max() / s,
                                                np.abs(g - g32[name]).max() / s))
