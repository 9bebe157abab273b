#!/usr/bin/env python3
"""Where does the run-to-run difference of the packed-FMA build first appear?  ONE xg_step_fwd from the same state, REPS times:
which of h1', c1', h2', c2' and of the attention weights alpha differ from the first repetition, in how many rows / elements.
XG_LIBRARY=.../libxgate_hip_pkfma.so python tools/r6/pkfma_where.py [precision] [rows] [reps]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import paramgen as pg
from tests.util import CFG, make_model, to_dev
from controllable_xgating_amd import _native as nv
from controllable_xgating_amd.model import _stream, _ws_ptr
def explain_context_element(ws_first, ws_cur, V, alpha, B, K, R):
    """The attention context (unnormalised, (B,R) followed by the B denominators) lives in the step workspace: find it by value, then say
    what the differing element is made of: which frame's term (or which wrong weight) accounts for the difference."""
    f0, f1 = ws_first.view(torch.float32).cpu().numpy(), ws_cur.view(torch.float32).cpu().numpy()
    ctx = torch.einsum("bk,bkr->br", alpha.double(), V.double()).cpu().numpy()          # normalised context, f64
    c0 = int(np.argmax(np.abs(ctx[0, :8])))
    n = (len(f0) - 8) // 4 * 4
    cand = np.arange(0, n, 4)
    ok = f0[cand + c0] != 0
    with np.errstate(all="ignore"):
        for i in range(8):
            ok &= np.abs(f0[cand + i].astype(np.float64) * ctx[0, c0] - f0[cand + c0].astype(np.float64) * ctx[0, i]) \
                  <= 1e-4 * np.abs(f0[cand + c0].astype(np.float64) * ctx[0, c0])
    offs = cand[ok]
    offs = [int(o) for o in offs if (f0[o:o + B * R] != f1[o:o + B * R]).any()]
    if not offs:
        print("  context buffer not located"); return
    o = offs[0]                                              # (a second match is the same rows' copy kept for the backward pass)
    den = np.ones(B)                                         # (the first match is the normalised context: weights = alpha)
    idx = np.nonzero(f0[o:o + B * R] != f1[o:o + B * R])[0]
    print("  context buffer at float %d; %d differing context elements" % (o, len(idx)))
    ex = alpha.double().cpu().numpy() * den[:, None]                                        # unnormalised weights (to rounding)
    Vn = V.double().cpu().numpy()
    lanes = sorted(set(int(j) % R // 2 % 64 for j in idx))
    print("  lanes of the differing elements (column / 2 mod 64):", lanes, "| columns all even:", all(int(j) % 2 == 0 for j in idx))
    for j in idx[:3]:
        b, c = divmod(int(j), R)
        a0, a1 = float(f0[o + j]), float(f1[o + j])
        terms = ex[b] * Vn[b, :, c]
        print("  video %d column %d (%s lane): first %.9g, now %.9g, difference %.6g" % (b, c, "low" if c % 2 == 0 else "high", a0, a1, a1 - a0))
        for which, (x, y) in (("first", (a1, a0)), ("now", (a0, a1))):
            # is y = x with one frame's term missing / doubled, or one frame's weight replaced by another frame's weight?
            dl = y - x
            k = int(np.argmin(np.abs(terms - (-dl)))); e = abs(terms[k] + dl)
            k2 = int(np.argmin(np.abs(terms - dl))); e2 = abs(terms[k2] - dl)
            sw = np.abs((ex[b][None, :] - ex[b][:, None]) * Vn[b, :, c][:, None] - dl)      # [k, j]: frame k used weight j instead of its own
            ks, js = np.unravel_index(np.argmin(sw), sw.shape)
            print("    if '%s' is the wrong one: drop frame %d -> residual %.3g | frame %d twice -> %.3g | frame %d with frame %d's weight -> %.3g   (ulp of the sum %.3g)"
                  % (which, k, e, k2, e2, ks, js, sw[ks, js], np.spacing(np.float32(abs(a0)))))


precision = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 128
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 20
d = pg.make_dims(**dict(CFG["c1"], B=rows))
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
    first = None
    explained = 0
    names = ["h1'", "c1'", "h2'", "c2'"]
    for rep in range(REPS):
        s = state0.clone()
        alpha = torch.zeros(d.B, d.K, device="cuda")
        nv.check(nv.lib().xg_step_fwd(_stream(), C.byref(dd), C.byref(ps), nv.ptr(tok), None, nv.ptr(V), nv.ptr(vproj),
                                      nv.ptr(x["pos_feats"]), C.byref(run), 0, wp, wn, nv.ptr(s), None, nv.ptr(alpha)), "xg_step_fwd")
        torch.cuda.synchronize()
        cur = (s.clone().view(4, d.B, -1), alpha.clone(), ws.clone())
        if first is None:
            first = cur
            continue
        msg = []
        for i, nm in enumerate(names):
            df = (cur[0][i] != first[0][i])
            if df.any():
                msg.append("%s: %d elements in rows %s (max %.3g)" % (nm, int(df.sum()), sorted(set(df.nonzero()[:, 0].tolist()))[:8],
                                                                      float((cur[0][i] - first[0][i]).abs().max())))
        da = cur[1] != first[1]
        if da.any():
            msg.append("alpha: %d elements in rows %s" % (int(da.sum()), sorted(set(da.nonzero()[:, 0].tolist()))[:8]))
        print("rep %2d:" % rep, "; ".join(msg) if msg else "identical")
        if msg and explained < 8:
            explained += 1
            explain_context_element(first[2], cur[2], V, cur[1], d.B, d.K, model.rnn_size)

