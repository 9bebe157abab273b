#!/usr/bin/env python3
"""xg_step_fwd in gemm_mode 3 (split-bf16 step products) against gemm_mode 0 from the SAME state / V / vproj / packed tiles:
per-step difference and run-to-run reproducibility of each mode.  usage: step_mode_check.py [rows]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from controllable_xgating_amd import SAModel, make_opt, _native as nv      # noqa: E402
from controllable_xgating_amd.model import _stream, _ws_ptr                 # noqa: E402
from oracle import paramgen as pg                                           # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 128
d = pg.make_dims(B=rows, K=26, R=512, A=1536, E=468, V=20000, C=14, L=20, F1=1536, F2=1024)
Pn = pg.make_params(d)
xn = pg.make_inputs(d, seed=0)
x = {k: torch.from_numpy(v).cuda() for k, v in xn.items()}
model = SAModel(make_opt(d, precision=os.environ.get("PREC_MODEL", "fp32")))
model.load_state_dict({k: torch.from_numpy(v) for k, v in Pn.items()}, strict=False)
model = model.cuda().eval()
with torch.no_grad():
    V = model.encode(x["feats_rgb"], x["feats_opfl"], x["feat_mask"])
    st = model.init_hidden(V, x["feat_mask"])
    state0 = torch.cat([st[0][0], st[0][1], st[1][0], st[1][1]], 0).contiguous()
    dd = model._dims(d.B, d.K, 1)
    ps, run0 = model._params_struct(), model._run(False)
    vproj = torch.empty(d.B, d.K, model.att_size, device="cuda")
    nv.check(nv.lib().xg_vproj(_stream(), C.byref(dd), C.byref(ps), nv.ptr(V), nv.ptr(vproj), C.byref(run0)), "xg_vproj")
    ws = model._pool.shared(dd, V.device)
    wp, wn = _ws_ptr(ws)
    tok = x["seq"][:, 1].contiguous()

    def step(mode, nsteps):
        run = model._run(False)
        run.gemm_mode = mode
        s = state0.clone()
        for _ in range(nsteps):
            nv.check(nv.lib().xg_step_fwd(_stream(), C.byref(dd), C.byref(ps), nv.ptr(tok), None, nv.ptr(V), nv.ptr(vproj),
                                          nv.ptr(x["pos_feats"]), C.byref(run), 0, wp, wn, nv.ptr(s), None, None), "xg_step_fwd")
            if os.environ.get("SYNC_EACH"):
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return s.cpu().numpy()
    if os.environ.get("WS_DIFF"):
        from tools.ws_map import ws_map, locate
        regs = ws_map(d.B, d.K, d.R, d.A, d.E, d.V, d.C, 128, 1)
        run = model._run(False); run.gemm_mode = int(os.environ.get("WS_MODE", "3"))
        def one(s):
            nv.check(nv.lib().xg_step_fwd(_stream(), C.byref(dd), C.byref(ps), nv.ptr(tok), None, nv.ptr(V), nv.ptr(vproj),
                                          nv.ptr(x["pos_feats"]), C.byref(run), 0, wp, wn, nv.ptr(s), None, None), "xg_step_fwd")
        s2 = state0.clone(); one(s2); one(s2); torch.cuda.synchronize()
        base_off = wp.value - ws.data_ptr()
        snaps = []
        for rep in range(12):
            s = s2.clone(); one(s); torch.cuda.synchronize()
            snaps.append((ws.cpu().numpy().copy(), s.cpu().numpy().copy()))
        ref_ws, ref_s = snaps[0]
        for rep, (w_, s_) in enumerate(snaps[1:], 1):
            df = np.flatnonzero(w_ != ref_ws)
            if len(df) == 0 and np.array_equal(s_, ref_s):
                continue
            words = sorted(set(((df - base_off) // 4).tolist()))
            names = {}
            for wd in words:
                nm, idx = locate(regs, wd * 4)
                names.setdefault(nm, []).append(idx)
            print("rep %d: state differs: %s; workspace regions: %s" % (rep, not np.array_equal(s_, ref_s),
                  {k: (len(v), v[:6]) for k, v in names.items()}))
            fa, fb = ref_ws.view(np.uint8)[base_off:].view(np.float32), w_.view(np.uint8)[base_off:].view(np.float32)
            for wd in words[:8]:
                print("    word %d (%s): %.9g vs %.9g" % (wd, locate(regs, wd * 4), fa[wd], fb[wd]))
        sys.exit(0)
    for nsteps in (1, 2, 3):
        ref = step(int(os.environ.get("REF_MODE", "0")), nsteps)
        for mode in [int(v) for v in os.environ.get("MODES", "0,3").split(",")]:
            worst, rep_worst = 0.0, 0.0
            first = None
            for rep in range(20):
                o = step(mode, nsteps)
                if first is None:
                    first = o
                rep_worst = max(rep_worst, float(np.abs(o - first).max()))
                worst = max(worst, float(np.abs(o - ref).max()))
            rr = np.argwhere(np.abs(o - first) > 0)
            if len(rr):
                comp = rr[:, 0] // d.B
                print("   run-to-run differing elements: %d; by state component (h1,c1,h2,c2): %s; rows %s; cols min/max %d %d; sample %s" % (
                    len(rr), np.bincount(comp, minlength=4).tolist(), sorted(set((rr[:, 0] % d.B).tolist()))[:12], rr[:, 1].min(), rr[:, 1].max(), rr[:6].tolist()))
            bad = np.argwhere(np.abs(o - ref) > 5e-6)
            print("steps %d mode %d: max |mode - fp32| %.3g, run-to-run %.3g, elements > 5e-6: %d %s" %
                  (nsteps, mode, worst, rep_worst, len(bad), bad[:4].tolist()))
