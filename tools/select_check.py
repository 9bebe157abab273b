"""Rollout token choice over per-tile statistics (xg_heads.hip: vocab_part_kernel / vocab_part16_kernel + roll_select_kernel) against the older
(B, V) product + one-workgroup-per-row pass (XG_NO_FUSED_SELECT=1 of the -DXG_DIAG library): the same paired SCST rollout --
sampled half with a temperature, greedy half -- and a replay, on a vocabulary that is not a multiple of the tile width; tokens
must be identical, log-probs equal to fp32 round-off, and the gradients of the SCST loss through both must agree.
usage: select_check.py  (spawns itself once per variant)"""
import os, sys, subprocess, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_variant(out_path):
    import numpy as np, torch
    from oracle import paramgen as pg
    from controllable_xgating_amd import RewardCriterion, SAModel, make_opt
    res = {}
    for name, V, temp in (("v1004_t0.7", 1004, 0.7), ("v2000_t1", 2000, 1.0), ("v333_t1.3", 333, 1.3), ("v5004_t0.9", 5004, 0.9)):
        d = pg.make_dims(B=12, K=10, R=128, A=160, E=64, V=V, C=14, L=12, F1=96, F2=64)
        Pn = pg.make_params(d, logit_gain=1.0)
        x = {k: torch.from_numpy(v).cuda() for k, v in pg.make_inputs(d, seed=0).items()}
        u = torch.from_numpy(pg.uniform("uni_sel", (d.L + 1, d.B), 17)).cuda()
        reward = torch.from_numpy(pg.uniform("rew_sel", (d.B, 1), 5)).cuda() - 0.5
        model = SAModel(make_opt(d))
        model.load_state_dict({k: torch.from_numpy(v) for k, v in Pn.items()}, strict=False)
        model = model.cuda(); model.train()
        gen, slp, greedy, n = model.sample_pair(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"],
                                                {"sample_max": 0, "uniforms": u, "temperature": temp})
        loss = RewardCriterion()(slp, gen, reward.expand(-1, gen.shape[1]))
        loss.backward()
        with torch.no_grad():
            rseq, rslp, _ = model.sample(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"],
                                         {"forced_tokens": gen, "bn_update": False, "async": True})
        torch.cuda.synchronize()
        res[name] = dict(gen=gen.cpu().numpy(), slp=slp.detach().cpu().numpy(), greedy=greedy.cpu().numpy(), loss=float(loss.item()),
                         rseq=rseq.cpu().numpy(), rslp=rslp.cpu().numpy(),
                         gnorm={n: float(q.grad.norm().item()) for n, q in model.named_parameters() if q.grad is not None})
    pickle.dump(res, open(out_path, "wb"))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run_variant(sys.argv[1])
        sys.exit(0)
    import numpy as np, tempfile
    from controllable_xgating_amd import _native as nv
    # 32-column statistics, 80-column statistics (16 x 16 x 4 MFMA tiles), the production rule; the last one is the reference
    variants = ({"XG_VOCAB_TW": "32"}, {"XG_VOCAB_TW": "80"}, {}, {"XG_NO_FUSED_SELECT": "1"})
    outs = []
    for env_extra in variants:
        f = tempfile.mktemp(suffix=".pkl")
        env = dict(os.environ, XG_LIBRARY=nv.LIB_DIAG_PATH, **env_extra)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), f], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stdout[-2000:], r.stderr[-3000:]); sys.exit(1)
        outs.append(pickle.load(open(f, "rb"))); os.remove(f)
    bad = []
    ref = outs[-1]
    for vi, got in enumerate(outs[:-1]):
        print("variant", variants[vi] or "production rule")
        for name in got:
            p, q = got[name], ref[name]
            if not np.array_equal(p["gen"], q["gen"]) or not np.array_equal(p["greedy"], q["greedy"]) or not np.array_equal(p["rseq"], q["rseq"]):
                bad.append((vi, name, "tokens differ", int((p["gen"] != q["gen"]).sum()), int((p["greedy"] != q["greedy"]).sum())))
                continue
            e1 = float(np.abs(p["slp"] - q["slp"]).max()); e2 = float(np.abs(p["rslp"] - q["rslp"]).max())
            if e1 > 2e-5 or e2 > 2e-5 or abs(p["loss"] - q["loss"]) > 1e-5 * max(1.0, abs(q["loss"])):
                bad.append((vi, name, "log-probs", e1, e2, p["loss"], q["loss"]))
            for n, g in p["gnorm"].items():
                if abs(g - q["gnorm"][n]) > 1e-4 * max(1e-6, q["gnorm"][n]) + 1e-9:
                    bad.append((vi, name, n, g, q["gnorm"][n]))
            print("  ", name, "tokens identical (%d sampled, %d greedy), max |dlogp| %.1e / %.1e (replay)" % (p["gen"].size, p["greedy"].size, e1, e2))
    print("bad", bad)
    sys.exit(1 if bad else 0)
