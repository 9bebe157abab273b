"""Quick check of the dataflow step kernel: xg_step_fwd (in-place state, rollout form) against the three-launch form
(XG_NO_DSTEP=1 in the -DXG_DIAG library) on the same inputs, then its average duration."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import torch


def run(nrep):
    import bench
    from controllable_xgating_amd import SAModel, make_opt, _native as nv
    from controllable_xgating_amd.model import _stream, _ws_ptr
    B = int(os.environ.get("DS_B", "128"))
    dev = torch.device("cuda", 0)
    model = SAModel(make_opt(None)).to(dev); model.eval()
    x = bench.synth_inputs(B, 26, 20, 20000, 512, 1536, 1024, 14, 0, dev)
    with torch.no_grad():
        V = model.encode(x["feats_rgb"], x["feats_opfl"], x["feat_mask"])
        st = model.init_hidden(V, x["feat_mask"])
        state = torch.cat([st[0][0], st[0][1], st[1][0], st[1][1]], 0).contiguous()
        d = model._dims(B, 26, 1)
        ps, run_ = model._params_struct(), model._run(False)
        vproj = torch.empty(B, 26, model.att_size, device=dev)
        nv.check(nv.lib().xg_vproj(_stream(), C.byref(d), C.byref(ps), nv.ptr(V), nv.ptr(vproj), C.byref(run_)), "xg_vproj")
        ws = model._pool.shared(d, dev)
        wp, wn = _ws_ptr(ws)
        pos = x["pos_feats"].contiguous()
        alpha = torch.zeros(B, 26, device=dev)
        outs = []
        s0 = state.clone()
        for t in range(3):
            tok = x["seq"][:, t].contiguous()
            nv.check(nv.lib().xg_step_fwd(_stream(), C.byref(d), C.byref(ps), nv.ptr(tok), None, nv.ptr(V), nv.ptr(vproj),
                                          nv.ptr(pos), C.byref(run_), t, wp, wn, nv.ptr(s0), None, nv.ptr(alpha)), "xg_step_fwd")
            torch.cuda.synchronize()
            outs.append((s0.clone().cpu(), alpha.clone().cpu()))
        tok = x["seq"][:, 1].contiguous()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(10):
            nv.lib().xg_step_fwd(_stream(), C.byref(d), C.byref(ps), nv.ptr(tok), None, nv.ptr(V), nv.ptr(vproj), nv.ptr(pos),
                                 C.byref(run_), 0, wp, wn, nv.ptr(state), None, None)
        torch.cuda.synchronize(); e0.record()
        for _ in range(nrep):
            nv.lib().xg_step_fwd(_stream(), C.byref(d), C.byref(ps), nv.ptr(tok), None, nv.ptr(V), nv.ptr(vproj), nv.ptr(pos),
                                 C.byref(run_), 0, wp, wn, nv.ptr(state), None, None)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / nrep
        if hasattr(nv.lib(), "xg_debug_dstep_err"):        # (diag library) a consumer's bounded spin gave up somewhere: wrong results
            flag = C.c_int(0)
            rc = nv.lib().xg_debug_dstep_err(_stream(), C.byref(d), wp, wn, C.byref(flag))
            if rc != 0 or flag.value != 0:
                raise SystemExit("dataflow step kernel: time-out flag %d (rc %d)" % (flag.value, rc))
    return outs, us


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        outs, us = run(200)
        torch.save({"outs": outs, "us": us}, sys.argv[2])
    else:
        from controllable_xgating_amd import _native as nv
        res = {}
        variants = [("dstep", {"XG_LIBRARY": nv.LIB_DIAG_PATH, "XG_DSTEP": "1"}), ("three_launch", {})]
        for v in os.environ.get("DS_VARIANTS", "").split(";"):       # e.g. DS_VARIANTS="s1first:XG_S1_FIRST=1;v2:XG_LIBRARY=/path/lib.so"
            if v:
                nm, kvs = v.split(":", 1)                              # name:KEY=VAL[,KEY=VAL...]; XG_LIBRARY may be given
                env = dict(XG_LIBRARY=nv.LIB_DIAG_PATH)
                env.update(kv.split("=", 1) for kv in kvs.split(","))
                variants.append((nm, env))
        for name, env in variants:
            f = "/tmp/dstep_%s.pt" % name
            r = subprocess.run(["timeout", "120", sys.executable, os.path.abspath(__file__), "child", f], env=dict(os.environ, **env),
                               capture_output=True, text=True)
            if r.returncode != 0:
                print(name, "FAILED rc", r.returncode, r.stderr[-1500:]); sys.exit(1)
            res[name] = torch.load(f)
        a, b = res["dstep"], res["three_launch"]
        for t, ((sa, aa), (sb, ab)) in enumerate(zip(a["outs"], b["outs"])):
            print("step %d: max |state diff| %.3g (scale %.3g)  max |alpha diff| %.3g" % (t, float((sa - sb).abs().max()), float(sb.abs().max()),
                                                                                         float((aa - ab).abs().max())))
        worst = max(max(float((sa - sb).abs().max()), float((aa - ab).abs().max())) for (sa, aa), (sb, ab) in zip(a["outs"], b["outs"]))
        print("us per step: " + "   ".join("%s %.2f" % (k, v["us"]) for k, v in res.items()))
        if worst > 1e-5:
            print("MISMATCH %.3g" % worst); sys.exit(2)
        for k, v in res.items():
            if k not in ("dstep", "three_launch"):
                print("%s vs three_launch: max |state diff| %.3g" % (k, max(float((x[0] - y[0]).abs().max()) for x, y in zip(v["outs"], b["outs"]))))
