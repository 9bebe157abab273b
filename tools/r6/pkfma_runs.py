#!/usr/bin/env python3
"""Failure table of the in-kernel reproduction of the packed-FMA fault (docs/pkfma_hazard.md): three consecutive xg_step_fwd calls on an
in-place state, REPS times per shape; counts the repetitions whose final state differs from the first one, and where.
XG_LIBRARY=controllable_xgating_amd/lib/libxgate_hip_<variant>.so python tools/r6/pkfma_runs.py [reps]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import paramgen as pg
from tests.util import CFG, make_model, to_dev
from controllable_xgating_amd import _native as nv
from controllable_xgating_amd.model import _stream, _ws_ptr
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
shapes = [("fp32", "c1", 128), ("bf16x3", "c1", 128), ("bf16", "c1", 128), ("fp32", "c1", 64), ("bf16x3", "c1", 64), ("fp32", "c1", 40),
          ("bf16", "c5", 128), ("bf16x3", "c5", 128), ("fp32", "c5", 64)]
print("library:", os.environ.get("XG_LIBRARY", "product"), "| repetitions per shape:", REPS)
for precision, cfg, rows in shapes:
    d = pg.make_dims(**dict(CFG[cfg], B=rows))
    x = to_dev(pg.make_inputs(d, seed=0))
    model = make_model(d, train=False, precision=precision)
    with torch.no_grad():
        V = model.encode(x["feats_rgb"], x["feats_opfl"], x["feat_mask"])
        st = model.init_hidden(V, x["feat_mask"])
        state0 = torch.cat([st[0][0], st[0][1], st[1][0], st[1][1]], 0).contiguous()
        dd = model._dims(d.B, d.K, 1)
        ps, run = model._params_struct(), model._run(False)
        vproj = torch.empty(d.B, d.K, model.att_size, device="cuda")
        nv.check(nv.lib().xg_vproj(_stream(), C.byref(dd), C.byref(ps), nv.ptr(V), nv.ptr(vproj), C.byref(run)), "xg_vproj")
        ws = model._pool.shared(dd, V.device)
        wp, wn = _ws_ptr(ws)
        tok = x["seq"][:, 1].contiguous()
        first, bad, worst, nel = None, 0, 0.0, 0
        for rep in range(REPS):
            s = state0.clone()
            for _ in range(3):
                nv.check(nv.lib().xg_step_fwd(_stream(), C.byref(dd), C.byref(ps), nv.ptr(tok), None, nv.ptr(V), nv.ptr(vproj),
                                              nv.ptr(x["pos_feats"]), C.byref(run), 0, wp, wn, nv.ptr(s), None, None), "xg_step_fwd")
            torch.cuda.synchronize()
            if first is None:
                first = s.clone()
            elif not torch.equal(s, first):
                bad += 1
                diff = (s - first).abs()
                worst = max(worst, float(diff.max())); nel = max(nel, int((diff > 0).sum()))
    print("%-7s %-3s %4d rows: %2d of %d repetitions differ from the first%s" % (
        precision, cfg, rows, bad, REPS - 1, "" if not bad else " (largest difference %.3g, up to %d state elements)" % (worst, nel)))
    del model
