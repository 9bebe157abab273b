import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from util import CFG, to_dev
from oracle import paramgen as pg
import test_gpu_parity as T
for precision in ("bf16x3", "bf16"):
    d = pg.make_dims(**dict(CFG["c1"], B=64))
    x = to_dev(pg.make_inputs(d, seed=0))
    model = T.make_model(d, train=False, precision=precision)
    with torch.no_grad():
        Vs = [model.encode(x["feats_rgb"], x["feats_opfl"], x["feat_mask"]).clone() for _ in range(4)]
        print(precision, "encode equal:", [bool(torch.equal(Vs[0], v)) for v in Vs[1:]], [float((Vs[0]-v).abs().max()) for v in Vs[1:]])
        outs = [model.sample(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], {"sample_max": 1}) for _ in range(4)]
        print(precision, "seq equal:", [bool(torch.equal(outs[0][0], o[0])) for o in outs[1:]], "lp equal:", [bool(torch.equal(outs[0][1], o[1])) for o in outs[1:]],
              "max lp diff", [float((outs[0][1]-o[1]).abs().max()) for o in outs[1:]])
        # first differing step
        for o in outs[1:]:
            df = (outs[0][1] != o[1]).any(0).nonzero().flatten().tolist()
            print("   steps with differing logp:", df[:10], "rows differing at first such step:", int((outs[0][1][:, df[0]] != o[1][:, df[0]]).sum()) if df else 0)
