"""Host-side mirror of the reference's model surface (caption_src/SAModel.py:13-267) on top of the
HIP C ABI (include/xgate.h).  Same class / method / attribute names, argument meaning, return
shapes and ``state_dict`` keys as the reference, so ``starttrain.py`` / ``eval_utils.py`` style
drivers run unchanged:

    model = SAModel(opt); model.cuda(); model.train()
    logp, cat_logp = model(feats_rgb, feats_opfl, feat_mask, pos_feats, seq, seq_mask)
    loss = LanguageModelCriterion()(logp, seq, seq_mask); loss.backward()
    seq, seqLogprobs = model.sample(feats_rgb, feats_opfl, feat_mask, pos_feats, {'sample_max': 0})

PyTorch is plumbing here: tensor allocation, the stream, autograd bookkeeping.  All arithmetic
runs in hand-written gfx950 kernels behind ``libxgate_hip.so``; the nn.Linear / nn.LSTMCell /
nn.BatchNorm1d sub-modules below are parameter CONTAINERS only (they give the reference's
``state_dict`` names, shapes and default initialisation) and are never called.
"""
from __future__ import annotations

import argparse
import ctypes as C

import torch
import torch.nn as nn

from . import _native as nv


def make_opt(d=None, **kw):
    """argparse-style namespace with the flags SAModel reads (caption_src/myopts.py:3-88)."""
    base = dict(seed=1024, vocab_size=20000, category_size=14, input_encoding_size=468, rnn_size=512,
                num_layers=1, drop_prob_lm=0.0, seq_length=20, feat_size=1536, feat_size2=1024,
                att_size=1536, fusion_activity="ReLU")
    if d is not None:
        base.update(vocab_size=d.V, category_size=d.C, input_encoding_size=d.E, rnn_size=d.R,
                    seq_length=d.L, feat_size=d.F1, feat_size2=d.F2, att_size=d.A)
    base.update(kw)
    return argparse.Namespace(**base)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _with_grad_event(model, run):
    """XgRun for ONE backward call: carries the "decoder-side gradients are final" event (XgRun.grad_event) when a
    data-parallel helper (train.GradSync) has armed one on the model; otherwise the forward's XgRun as it is."""
    ev = getattr(model, "_grad_event", None)
    if ev is None:
        return run
    r = nv.XgRun()
    C.memmove(C.byref(r), C.byref(run), C.sizeof(nv.XgRun))
    r.grad_event = ev.cuda_event
    ev_head = getattr(model, "_grad_event_head", None)
    if ev_head is not None:
        r.grad_event_head = ev_head.cuda_event
    return r


# side-stream handles of the library (xg_aux_create: two HIP streams + events each), ONE per (device, caller stream) for the
# whole process and never destroyed: every model on that stream shares it.  A handle per model would add two more HIP streams
# per model, and streams beyond the first handful share the hardware's compute pipes with the earlier ones -- see
# train.shared_stream for what that costs.
_AUX_HANDLES = {}


class _Holder(nn.Module):
    """Bare container so parameter names nest like the reference's sub-modules."""


class _WorkspacePool:
    """Caller-owned workspaces (xg_workspace_bytes).  A forward that saves activations for a
    backward keeps its workspace until the backward has run; everything else shares scratch."""

    def __init__(self, gemm_mode=0):
        self.free = {}
        self.scratch = {}
        self.gemm_mode = gemm_mode            # sizes the workspaces: only the bf16 mode carries the bf16 mirror region (+50 %)

    @staticmethod
    def _key(dims, device):
        return (tuple(getattr(dims, f[0]) for f in dims._fields_), str(device))

    def _alloc(self, dims, device):
        n = nv.lib().xg_workspace_bytes_mode(C.byref(dims), self.gemm_mode)
        if n == 0:
            raise nv.XgError("xg_workspace_bytes_mode: invalid dims")
        return torch.zeros(n + 256, dtype=torch.uint8, device=device)     # zero-filled at first use: include/xgate.h

    def take(self, dims, device):
        lst = self.free.setdefault(self._key(dims, device), [])
        return lst.pop() if lst else self._alloc(dims, device)

    def give(self, dims, device, ws):
        self.free.setdefault(self._key(dims, device), []).append(ws)

    def shared(self, dims, device):
        k = self._key(dims, device)
        if k not in self.scratch:
            self.scratch[k] = self._alloc(dims, device)
        return self.scratch[k]


def _ws_ptr(ws):
    p = ws.data_ptr()
    p = (p + 255) & ~255
    return C.c_void_p(p), C.c_size_t(ws.numel() - 256)


class SAModel(nn.Module):
    """Drop-in for reference caption_src/SAModel.py:SAModel (CaptionModel(nn.Module))."""

    def __init__(self, opt):
        super().__init__()
        nv.lib()  # fail loudly, right here, if the HIP library is missing
        seed = opt.seed
        torch.manual_seed(seed)
        self.vocab_size = opt.vocab_size
        self.category_size = opt.category_size
        self.input_encoding_size = opt.input_encoding_size
        self.rnn_size = opt.rnn_size
        self.visual_size = opt.rnn_size
        self.num_layers = opt.num_layers
        self.drop_prob_lm = opt.drop_prob_lm
        self.seq_length = opt.seq_length
        self.att_size = opt.att_size
        self.feat_size, self.feat_size2 = opt.feat_size, opt.feat_size2
        self.ss_prob = 0.0
        self.done_beams = []
        R, E, A, p = opt.rnn_size, opt.input_encoding_size, opt.att_size, opt.drop_prob_lm

        def gate(src, tgt):  # reference sub_modules.py:18-31 (re-seeds, so siblings start identical)
            torch.manual_seed(seed)
            h = _Holder()
            h.gate = nn.Sequential(nn.Linear(src, tgt), nn.ReLU(), nn.Dropout(p))
            return h

        # construction order mirrors the reference so torch's default init consumes the RNG identically
        enc = _Holder()
        torch.manual_seed(seed)                                                    # sub_modules.py:86
        enc.visual_emb_rgb = nn.Sequential(nn.Linear(opt.feat_size, R), nn.BatchNorm1d(R), nn.ReLU(True))
        enc.visual_emb_opfl = nn.Sequential(nn.Linear(opt.feat_size2, R), nn.BatchNorm1d(R), nn.ReLU(True))
        enc.drop_out = nn.Dropout(p)
        enc.lstmcell_rgb = nn.LSTMCell(R, R)
        enc.lstmcell_opfl = nn.LSTMCell(R, R)
        enc.gate_rgb = gate(R, R)
        enc.gate_opfl = gate(R, R)
        torch.manual_seed(seed)                                                    # Fusion, sub_modules.py:55
        enc.fusion = _Holder()
        enc.fusion.late_fusion = nn.Sequential(nn.Linear(2 * R, R), getattr(nn, opt.fusion_activity)(), nn.Dropout(p))
        self.two_spatial_encoder = enc
        self.img_embed_h_1 = nn.Linear(R, R)
        self.img_embed_c_1 = nn.Linear(R, R)
        self.img_embed_h_2 = nn.Linear(R, R)
        self.img_embed_c_2 = nn.Linear(R, R)
        core = _Holder()
        torch.manual_seed(seed)                                                    # sub_modules.py:648
        core.gate = gate(E, R)

        def cell(in1):
            c = _Holder()
            c.i2h, c.a2h, c.h2h = nn.Linear(in1, 4 * R), nn.Linear(R, 4 * R), nn.Linear(R, 4 * R)
            c.dropout = nn.Dropout(p)
            return c

        core.lstm_1 = cell(E)
        core.lstm_2 = cell(R)
        core.dropout = nn.Dropout(p)
        core.v2a = nn.Linear(R, A)
        core.h2a = nn.Linear(2 * R, A)
        core.a2w = nn.Linear(A, 1)
        self.lstmcore = core
        self.embed = nn.Embedding(self.vocab_size, E)
        self.logit = nn.Linear(R, self.vocab_size)
        self.classifer = nn.Sequential(nn.Linear(R, 128), nn.ReLU(), nn.Dropout(p), nn.Linear(128, self.category_size))
        self.init_weights()
        if getattr(opt, "fusion_activity", "ReLU") != "ReLU":
            raise ValueError("only fusion_activity='ReLU' (the shipped recipe) is implemented in HIP")
        self._pool = _WorkspacePool({"fp32": 0, "bf16": 1, "bf16x3": 3}.get(getattr(opt, "precision", "fp32"), 1))
        self._flat = None
        self._ps_cache = None
        self._gs_cache = None
        self._offsets = []
        self._call = 0
        self.dropout_seed = None        # fixed seed of the dropout hash (tests / bench parity leg: a checker can regenerate the same masks); None = a fresh seed per call
        self._packed = None          # recurrent weights in MFMA-fragment order (xg_pack_weights), refreshed lazily
        self._packed_key = None
        self._packed_epoch = 0
        # arithmetic of the large GEMMs: 'fp32' (exact fp32 MFMA), 'bf16x3' (split-bf16, fp32-class accuracy, faster),
        # 'bf16' (bf16 operands / fp32 accumulate, for the large AND the per-step products: BASELINE.json configs[4]).
        # 'bf16x3' splits the per-step products too (three bf16 planes of the packed fp32 tiles, xg_step.hip PREC 2).
        self.precision = getattr(opt, "precision", "fp32")
        if self.precision not in ("fp32", "bf16", "bf16x3"):
            raise ValueError("precision must be 'fp32', 'bf16x3' or 'bf16'")

    def init_weights(self):  # SAModel.py:52-56
        initrange = 0.1
        self.embed.weight.data.uniform_(-initrange, initrange)
        self.logit.bias.data.fill_(0)
        self.logit.weight.data.uniform_(-initrange, initrange)

    # ------------------------------------------------------------------ plumbing
    def _named(self):
        return dict(self.named_parameters())

    def _plist(self):
        """The nn.Parameter objects in the C ABI's order (= state_dict order).  The objects themselves are stable for the
        life of the module (.cuda()/.to()/load_state_dict replace their DATA), so the walk over named_parameters() is
        done once."""
        pl = self.__dict__.get("_plist_cache")
        if pl is None:
            named = self._named()
            pl = [named[n] for n in nv.PARAM_NAMES]
            self.__dict__["_plist_cache"] = pl
        return pl

    def _param_list(self):
        return self._plist()

    def _ensure_flat(self):
        """All parameters live in ONE flat fp32 buffer (one RCCL all-reduce, one Adam launch);
        the nn.Parameters are views into it.  Rebuilt if .cuda()/.to() replaced the storages."""
        names = nv.PARAM_NAMES
        plist = self._plist()
        first = plist[0]
        if self._flat is not None and self._flat.device == first.device:
            base, ok = self._flat.data_ptr(), True
            for p, off in zip(plist, self._offsets):
                if p.data_ptr() != base + 4 * off:
                    ok = False
                    break
            if ok:
                return
        if not first.is_cuda:
            raise nv.XgError("SAModel runs on an MI355X only: call model.cuda() first (no CPU path)")
        total = sum((p.numel() + 63) // 64 * 64 for p in plist)
        flat = torch.zeros(total, dtype=torch.float32, device=first.device)
        gflat = torch.zeros(total, dtype=torch.float32, device=first.device)
        off = 0
        self._slices = {}
        self._offsets = []
        for n, p in zip(names, plist):
            if p.dtype != torch.float32:
                raise nv.XgError("fp32 parameters only")
            k = p.numel()
            v = flat[off:off + k].view_as(p)
            v.copy_(p.data)
            had_grad = p.grad is not None
            if had_grad:
                gflat[off:off + k].view_as(p).copy_(p.grad)
            p.data = v
            p.grad = gflat[off:off + k].view_as(p) if had_grad else None
            self._slices[n] = (off, k)
            self._offsets.append(off)
            off += (k + 63) // 64 * 64
        self._flat, self._gflat = flat, gflat
        self._ps_cache = None
        self._gs_cache = None

    def flat_parameters(self):
        self._ensure_flat()
        return self._flat

    def flat_grads(self):
        """Flat gradient buffer; binds every p.grad to its slice (zeroing is the caller's zero_grad)."""
        self._ensure_flat()
        base = self._gflat.data_ptr()
        for p, off in zip(self._plist(), self._offsets):
            if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                v = self._gflat[off:off + p.numel()].view_as(p)
                if p.grad is not None:
                    v.copy_(p.grad)
                else:
                    v.zero_()
                p.grad = v
        return self._gflat

    def _params_struct(self):
        """XgParams of the flat buffer's slices (cached until the buffer is rebuilt)."""
        self._ensure_flat()
        if self._ps_cache is None:
            ps = nv.XgParams()
            base = self._flat.data_ptr()
            for i, off in enumerate(self._offsets):
                setattr(ps, "p%d" % i, base + 4 * off)
            self._ps_cache = ps
        return self._ps_cache

    def _bn_struct(self):
        e = self.two_spatial_encoder
        s = nv.XgBnState()
        s.rgb_mean, s.rgb_var = e.visual_emb_rgb[1].running_mean.data_ptr(), e.visual_emb_rgb[1].running_var.data_ptr()
        s.opfl_mean, s.opfl_var = e.visual_emb_opfl[1].running_mean.data_ptr(), e.visual_emb_opfl[1].running_var.data_ptr()
        return s

    def _dims(self, B, K, T):
        d = nv.XgDims()
        d.B, d.K, d.R, d.A, d.E, d.V = B, K, self.rnn_size, self.att_size, self.input_encoding_size, self.vocab_size
        d.C, d.H, d.F1, d.F2, d.T = self.category_size, 128, self.feat_size, self.feat_size2, T
        return d

    def mark_params_changed(self, only_encoder_since_pack_early=False):
        """Tell the model its parameters were rewritten behind torch's back (a HIP kernel on the flat buffer, e.g.
        train.ClipAdam.step): the packed shadow of the recurrent weights is rebuilt before the next call.  Updates made
        through torch on the parameters themselves (optimizers, load_state_dict, ``p.copy_()``, ``p.mul_()`` under
        no_grad) are noticed by themselves (tensor version counters).  NOT noticed: in-place writes through ``p.data``
        (``p.data.copy_()``, ``p.data.uniform_()`` -- common in code written against the reference's torch 0.3): ``.data``
        has its own version counter.  Call this method after such writes.
        ``only_encoder_since_pack_early=True`` is the statement of an optimizer that called ``pack_early()`` right behind its
        update of every parameter group except the CG encoder's and has written NOTHING but the encoder's segment of the flat
        buffer since (train.ClipAdam): only then may the next call keep the early-packed decoder tiles.  Any other caller gets
        the full re-pack."""
        self._packed_epoch += 1
        if not only_encoder_since_pack_early:
            self._early_key = None           # whatever pack_early() covered may be stale again: the next call re-packs everything

    def _packed_dtype(self):
        """Element type of the packed recurrent weights (include/xgate.h: XgRun.packed_dtype): bf16 tiles for the bf16 arithmetic,
        three pre-split bf16 planes for split-bf16, fp32 tiles otherwise."""
        over = getattr(self, "_packed_dtype_override", None)      # tests: split-bf16 over plain fp32 tiles (split in registers)
        return {"bf16": 1, "bf16x3": 2}.get(self.precision, 0) if over is None else over

    def pack_early(self):
        """For an optimizer that has JUST updated every parameter group except the CG encoder's on the current stream (and will
        call mark_params_changed() when the rest is done): refresh the packed tiles of the decoder's matrices here and now, under
        the encoder's backward, instead of at the head of the next iteration (xg_pack_weights_part, part 1).  The next
        _packed_ptr() then only packs the encoder's tiles (and, with bf16 tiles, the bf16 copies of the encoder's weights).  No-op
        when there is nothing to gain (no shadow yet, a HIP-graph capture)."""
        if self._packed is None or torch.cuda.is_current_stream_capturing():
            return
        self._ensure_flat()
        dtype = self._packed_dtype()
        key = (self._flat.data_ptr(), self._flat._version, self._packed_epoch, dtype, tuple(p._version for p in self._plist()))
        d = self._dims(1, 1, 1)
        nbytes = nv.lib().xg_packed_bytes(C.byref(d), dtype)
        if nbytes == 0 or self._packed.numel() < nbytes + 16:
            return
        ptr = (self._packed.data_ptr() + 15) & ~15
        ps = self._params_struct()
        nv.check(nv.lib().xg_pack_weights_part(_stream(), C.byref(d), C.byref(ps), C.c_void_p(ptr), C.c_size_t(nbytes), dtype, 1, 1),
                 "xg_pack_weights_part")
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        # valid for exactly the NEXT epoch (the optimizer's mark_params_changed) with nothing else changed in between
        self._early_event, self._early_epoch, self._early_key = ev, self._packed_epoch + 1, key[:2] + key[3:]

    def _aux_handle(self):
        """Side-stream handle (include/xgate.h: xg_aux_create) for the current device and stream, created on first use."""
        if torch.cuda.is_current_stream_capturing():
            return None      # a HIP graph is being captured: one stream (forked streams crash or replay slowly, train.GraphedXEStep)
        st = torch.cuda.current_stream()
        key = (st.device.index, st.cuda_stream)
        h = _AUX_HANDLES.get(key)
        if h is None:
            out = C.c_void_p()
            nv.check(nv.lib().xg_aux_create(C.byref(out)), "xg_aux_create")
            h = _AUX_HANDLES[key] = out.value
        return h

    def _packed_ptr(self):
        """Device pointer of the packed recurrent weights (include/xgate.h: xg_pack_weights), valid for the current
        parameter values; None when the shapes do not allow it (rnn_size % 8 != 0)."""
        self._ensure_flat()
        dtype = self._packed_dtype()
        key = (self._flat.data_ptr(), self._flat._version, self._packed_epoch, dtype,
               tuple(p._version for p in self._plist()))
        if key != self._packed_key:
            d = self._dims(1, 1, 1)
            nbytes = nv.lib().xg_packed_bytes(C.byref(d), dtype)
            if nbytes == 0:
                self._packed = None
            else:
                fresh = self._packed is None or self._packed.device != self._flat.device or self._packed.numel() < nbytes + 16
                if fresh:
                    self._packed = torch.empty(nbytes + 16, dtype=torch.uint8, device=self._flat.device)
                ptr = (self._packed.data_ptr() + 15) & ~15
                ps = self._params_struct()
                # the decoder's matrices may already have been refreshed behind the optimizer's update of their parameter group
                # (pack_early, under the CG encoder's backward): then only the encoder's four tiles are left for the head of the
                # iteration, behind an event that completed long ago
                early = (not fresh and getattr(self, "_early_key", None) == key[:2] + key[3:] and
                         self._early_epoch == self._packed_epoch and not torch.cuda.is_current_stream_capturing())
                if early:
                    torch.cuda.current_stream().wait_event(self._early_event)
                    nv.check(nv.lib().xg_pack_weights_part(_stream(), C.byref(d), C.byref(ps), C.c_void_p(ptr), C.c_size_t(nbytes), dtype, 1, 2),
                             "xg_pack_weights_part")
                else:
                    nv.check(nv.lib().xg_pack_weights(_stream(), C.byref(d), C.byref(ps), C.c_void_p(ptr), C.c_size_t(nbytes), dtype, 1),
                             "xg_pack_weights")
                self._early_key = None
                # the shadow is (re)written on THIS stream: calls on any other stream must wait for it (a rollout on a side
                # stream right after an optimizer step, driver.scst_rollouts(mode="streams"))
                if torch.cuda.is_current_stream_capturing():       # (a HIP graph: ordering is the graph's own business)
                    self._packed_event = None
                else:
                    st = torch.cuda.current_stream()
                    ev = torch.cuda.Event()
                    ev.record(st)
                    self._packed_event, self._packed_stream = ev, (st.device.index, st.cuda_stream)
            self._packed_key = key
        if self._packed is None:
            return None
        ev = getattr(self, "_packed_event", None)
        if ev is not None:
            st = torch.cuda.current_stream()
            if (st.device.index, st.cuda_stream) != self._packed_stream:
                st.wait_event(ev)
        return (self._packed.data_ptr() + 15) & ~15

    def _run(self, save, seed=None):
        r = nv.XgRun()
        r.train = 1 if self.training else 0
        r.drop_p = float(self.drop_prob_lm)
        if seed is None and self.dropout_seed is not None:
            seed = int(self.dropout_seed) & 0xFFFFFFFF
        if seed is None:
            self._call += 1
            seed = (int(torch.initial_seed()) * 2654435761 + self._call * 40503) & 0xFFFFFFFF
        r.seed = seed
        r.save = 1 if save else 0
        r.bn_momentum, r.bn_eps = 0.1, 1e-5
        r.gemm_mode = {"fp32": 0, "bf16": 1, "bf16x3": 3}[self.precision]
        r.packed = self._packed_ptr()
        r.packed_dtype = self._packed_dtype()
        r.aux = self._aux_handle()
        pe = getattr(self, "_prof_events", None)       # measurement hook (bench.py): a pair of timing events around the T decoder steps
        if pe is not None:
            r.prof_event0, r.prof_event1 = pe[0].cuda_event, pe[1].cuda_event
        return r

    @staticmethod
    def _batch(feats_rgb, feats_opfl, feat_mask, pos_feats, seq=None, seq_mask=None):
        def f32(t):
            return t.detach().contiguous().float()
        keep = [f32(feats_rgb), f32(feats_opfl), f32(feat_mask), f32(pos_feats),
                None if seq is None else seq.detach().contiguous().long(),
                None if seq_mask is None else f32(seq_mask)]
        for t in keep:
            if t is not None and not t.is_cuda:
                raise nv.XgError("inputs must be CUDA (HIP) tensors")
        b = nv.XgBatch()
        b.feats_rgb, b.feats_opfl, b.feat_mask, b.pos_feats = (t.data_ptr() for t in keep[:4])
        b.seq = keep[4].data_ptr() if keep[4] is not None else None
        b.seq_mask = keep[5].data_ptr() if keep[5] is not None else None
        return b, keep

    def _bump_bn(self, n=1):
        if self.training:
            ts = [m.num_batches_tracked for m in (self.two_spatial_encoder.visual_emb_rgb[1], self.two_spatial_encoder.visual_emb_opfl[1])
                  if m.num_batches_tracked is not None]
            if ts:
                torch._foreach_add_(ts, n)          # (one launch for both counters)

    # ------------------------------------------------------------------ reference surface
    def forward(self, feats_rgb, feats_opfl, feat_mask, pos_feats, seq, seq_mask):
        """SAModel.forward (SAModel.py:67-115): (m,K,F) x2, (m,K), (m,R), (m,T) int64, (m,T) ->
        log-probs (m,T,V) and category log-probs (m,T,C).  With self.ss_prob > 0 in train mode the scheduled-sampling
        path (SAModel.py:89-99, xg_forward_ss) runs instead of the hoisted teacher-forced one."""
        params = self._param_list()
        save = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        if self.training and self.ss_prob > 0.0:                                 # scheduled sampling, SAModel.py:89-99
            T, B = seq.shape[1], seq.shape[0]
            u = getattr(self, "ss_uniforms", None)                               # test hook: (u_sel, u_tok), each (T,B)
            if u is None:
                u = (torch.rand(T, B, device=seq.device), torch.rand(T, B, device=seq.device))
            ss = (float(self.ss_prob), u[0].contiguous().float(), u[1].contiguous().float())
            return _XEFunction.apply(self, save, ss, feats_rgb, feats_opfl, feat_mask, pos_feats, seq, seq_mask, *params)
        return _XEFunction.apply(self, save, None, feats_rgb, feats_opfl, feat_mask, pos_feats, seq, seq_mask, *params)

    def xe_loss(self, feats_rgb, feats_opfl, feat_mask, pos_feats, seq, seq_mask, cap_classes=None,
                class_mask=None, weight_class=0.0):
        """Fused fast path: forward + LanguageModelCriterion (+ weight_class * ClassiferCriterion)
        without materialising the (m,T,V) log-prob tensor or its gradient
        (starttrain.py:125-129).  Returns the scalar loss tensor; .backward() works."""
        params = self._param_list()
        save = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        return _XELossFunction.apply(self, save, feats_rgb, feats_opfl, feat_mask, pos_feats, seq, seq_mask, cap_classes,
                                     class_mask, float(weight_class), *params)

    def sample_pair(self, feats_rgb, feats_opfl, feat_mask, pos_feats, opt={}):
        """The two rollouts of one SCST iteration in ONE batched pass: ``sample(..., {'sample_max': 0})``
        (starttrain.py:131) and the greedy baseline ``sample(..., {'sample_max': 1})`` (myutils.py:45-48).
        Returns (gen (m,L), sample_logprobs (m,L) [differentiable], greedy (m,L), n (2,) int32 device tensor: the
        reference's early-exit lengths of the two rollouts -- trim with them, or hand n[:1] to RewardCriterion).
        In train mode the BatchNorm running statistics end up exactly where the reference's TWO sample() calls leave
        them (two momentum updates with the unbiased N/(N-1) variance of the un-repeated batch).  With train-mode dropout
        (drop_prob_lm > 0) the two rollouts run as two calls with independent masks, like the reference's."""
        temperature = float(opt.get("temperature", 1.0))
        params = self._param_list()
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        if self.training and self.drop_prob_lm > 0.0:
            # Train-mode dropout: the reference's two sample() calls (starttrain.py:131, myutils.py:45) draw INDEPENDENT masks,
            # in the encoder and img_embed as well as in the decoder.  The one-batch form below runs the encoder once and would
            # hand both halves the same encoder realisation, so this case takes the reference's own shape: two rollouts, two
            # seeds (each with its BatchNorm update), the greedy one without saved activations.
            uni = opt.get("uniforms", None)
            gen, slp, n_s = _RolloutFunction.apply(self, feats_rgb, feats_opfl, feat_mask, pos_feats, nv.XG_ROLLOUT_SAMPLE, uni,
                                                   None, temperature, need_grad, True, *params)
            with torch.no_grad():
                greedy, _, n_g = _RolloutFunction.apply(self, feats_rgb, feats_opfl, feat_mask, pos_feats, nv.XG_ROLLOUT_GREEDY,
                                                        None, None, 1.0, False, True, *params)
            return gen, slp, greedy, torch.cat([n_s.reshape(1), n_g.reshape(1)])
        return _RolloutPairFunction.apply(self, feats_rgb, feats_opfl, feat_mask, pos_feats, opt.get("uniforms", None),
                                          temperature, need_grad, *params)

    def init_hidden(self, feat, feat_mask):
        """SAModel.init_hidden (SAModel.py:58-65) -> [(h1,c1),(h2,c2)], each (1,m,R)."""
        B, K, R = feat.shape
        d = self._dims(B, K, 1)
        ws = self._pool.shared(d, feat.device)
        wp, wn = _ws_ptr(ws)
        state = torch.empty(4, B, R, dtype=torch.float32, device=feat.device)
        fm = feat_mask.detach().contiguous().float()
        if fm.dim() == 1:
            fm = fm.unsqueeze(0).expand(B, K).contiguous()
        ps = self._params_struct()
        nv.check(nv.lib().xg_init_hidden(_stream(), C.byref(d), C.byref(ps), nv.ptr(feat.detach().contiguous().float()),
                                         nv.ptr(fm), wp, wn, nv.ptr(state)), "xg_init_hidden")
        return [(state[0:1], state[1:2]), (state[2:3], state[3:4])]

    def encode(self, feats_rgb, feats_opfl, feat_mask):
        """two_spatial_encoder(feats_rgb, feats_opfl, feat_mask) (sub_modules.py:118-159), no grad."""
        B, K, _ = feats_rgb.shape
        d = self._dims(B, K, 1)
        ws = self._pool.shared(d, feats_rgb.device)
        wp, wn = _ws_ptr(ws)
        b, keep = self._batch(feats_rgb, feats_opfl, feat_mask, feats_rgb.new_zeros(B, self.rnn_size))
        V = torch.empty(B, K, self.rnn_size, dtype=torch.float32, device=feats_rgb.device)
        ps, bn, run = self._params_struct(), self._bn_struct(), self._run(False)
        nv.check(nv.lib().xg_encoder_fwd(_stream(), C.byref(d), C.byref(ps), C.byref(bn), C.byref(b), C.byref(run),
                                         wp, wn, nv.ptr(V)), "xg_encoder_fwd")
        self._bump_bn()
        return V

    def get_logprobs_state(self, it, feats, pos_feats, state):
        """SAModel.get_logprobs_state (SAModel.py:117-127): one step, mask of ones."""
        B, K, R = feats.shape
        d = self._dims(B, K, 1)
        ws = self._pool.shared(d, feats.device)
        wp, wn = _ws_ptr(ws)
        feats = feats.detach().contiguous().float()
        st = torch.cat([state[0][0], state[0][1], state[1][0], state[1][1]], 0).contiguous().float()
        v2a = self.lstmcore.v2a
        key = (feats.data_ptr(), feats._version, B, K, v2a.weight._version, v2a.bias._version, self._packed_epoch, self.precision)
        if getattr(self, "_vproj_key", None) != key:
            self._vproj_cache = torch.empty(B, K, self.att_size, dtype=torch.float32, device=feats.device)
            ps, run0 = self._params_struct(), self._run(False)
            nv.check(nv.lib().xg_vproj(_stream(), C.byref(d), C.byref(ps), nv.ptr(feats), nv.ptr(self._vproj_cache), C.byref(run0)),
                     "xg_vproj")
            self._vproj_key = key
            self._vproj_feats = feats
        logp = torch.empty(B, self.vocab_size, dtype=torch.float32, device=feats.device)
        ps, run = self._params_struct(), self._run(False)
        pos = pos_feats.detach().contiguous().float()
        tok = it.detach().contiguous().long().to(feats.device)
        nv.check(nv.lib().xg_step_fwd(_stream(), C.byref(d), C.byref(ps), nv.ptr(tok), None, nv.ptr(feats),
                                      nv.ptr(self._vproj_cache), nv.ptr(pos), C.byref(run), 0, wp, wn, nv.ptr(st),
                                      nv.ptr(logp), None), "xg_step_fwd")
        return logp, [(st[0:1], st[1:2]), (st[2:3], st[3:4])]

    def sample(self, feats_rgb, feats_opfl, feat_mask, pos_feats, opt={}):
        """SAModel.sample (SAModel.py:163-219) -> seq (m,n) int64, seqLogprobs (m,n)."""
        sample_max = opt.get("sample_max", 1)
        beam_size = opt.get("beam_size", 1)
        temperature = opt.get("temperature", 1.0)
        if beam_size > 1:
            from .beam import sample_beam
            feats = self.encode(feats_rgb, feats_opfl, feat_mask)
            return sample_beam(self, feats, feat_mask, pos_feats, opt)
        mode = nv.XG_ROLLOUT_GREEDY if sample_max else nv.XG_ROLLOUT_SAMPLE
        forced = opt.get("forced_tokens", None)
        if forced is not None:
            mode = nv.XG_ROLLOUT_REPLAY
        need_grad = torch.is_grad_enabled() and mode != nv.XG_ROLLOUT_GREEDY and any(p.requires_grad for p in self.parameters())
        params = self._param_list()
        uniforms = opt.get("uniforms", None)
        bn_update = bool(opt.get("bn_update", True))
        seq, slp, n = _RolloutFunction.apply(self, feats_rgb, feats_opfl, feat_mask, pos_feats, mode, uniforms, forced,
                                             float(temperature), need_grad, bn_update, *params)
        if opt.get("async", False):            # no host sync: full-width (m, L) tensors + the device-side n
            return seq, slp, n
        n = int(n.item())                      # ONE host sync per rollout (the reference syncs every step, :206)
        return seq[:, :n], slp[:, :n]

    def sample_beam(self, feats, feat_masks, pos_feats, opt={}):
        from .beam import sample_beam
        return sample_beam(self, feats, feat_masks, pos_feats, opt)


# ====================================================================== autograd glue
def _grads_bound(model):
    base = model._gflat.data_ptr()
    for p, off in zip(model._plist(), model._offsets):
        g = p.grad
        if g is None or g.data_ptr() != base + 4 * off:
            return False
    return True


def _grads_struct(model, device):
    """XgParams struct of gradient pointers (the C ABI accumulates into them).  If every p.grad is
    already bound to the model's flat gradient buffer (model.flat_grads()), accumulate there directly
    and hand autograd nothing; otherwise use a fresh zero buffer and return views for autograd."""
    model._ensure_flat()
    model._grad_writes = getattr(model, "_grad_writes", 0) + 1     # a backward is about to add to the gradient buffer
    if _grads_bound(model):
        if model._gs_cache is None:
            s = nv.XgParams()
            base = model._gflat.data_ptr()
            for i, off in enumerate(model._offsets):
                setattr(s, "p%d" % i, base + 4 * off)
            model._gs_cache = s
        return None, model._gs_cache
    g = torch.zeros_like(model._flat)
    s = nv.XgParams()
    for i, off in enumerate(model._offsets):
        setattr(s, "p%d" % i, g.data_ptr() + 4 * off)
    return g, s


def _grad_views(model, g):
    if g is None:
        return [None] * len(nv.PARAM_NAMES)
    return [g[off:off + p.numel()].view_as(p) for p, off in zip(model._plist(), model._offsets)]


class _XEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, save, ss, feats_rgb, feats_opfl, feat_mask, pos_feats, seq, seq_mask, *params):
        B, K, _ = feats_rgb.shape
        T = seq.shape[1]
        dev = feats_rgb.device
        d = model._dims(B, K, T)
        ws = model._pool.take(d, dev) if save else model._pool.shared(d, dev)
        wp, wn = _ws_ptr(ws)
        b, keep = model._batch(feats_rgb, feats_opfl, feat_mask, pos_feats, seq, seq_mask)
        logp = torch.empty(B, T, model.vocab_size, dtype=torch.float32, device=dev)
        cat = torch.empty(B, T, model.category_size, dtype=torch.float32, device=dev)
        ps, bn, run = model._params_struct(), model._bn_struct(), model._run(save)
        if ss is None:
            nv.check(nv.lib().xg_forward_xe(_stream(), C.byref(d), C.byref(ps), C.byref(bn), C.byref(b), C.byref(run),
                                            wp, wn, nv.ptr(logp), nv.ptr(cat)), "xg_forward_xe")
        else:
            nv.check(nv.lib().xg_forward_ss(_stream(), C.byref(d), C.byref(ps), C.byref(bn), C.byref(b), C.byref(run),
                                            ss[0], nv.ptr(ss[1]), nv.ptr(ss[2]), wp, wn, nv.ptr(logp), nv.ptr(cat)),
                     "xg_forward_ss")
        model._bump_bn()
        ctx.model, ctx.d, ctx.ws, ctx.keep, ctx.run, ctx.saved_ws, ctx.ss = model, d, ws, keep, run, save, ss is not None
        return logp, cat

    @staticmethod
    def backward(ctx, dlogp, dcat):
        model, d = ctx.model, ctx.d
        if not ctx.saved_ws:
            raise nv.XgError("backward without saved activations")
        dev = ctx.keep[0].device
        g, gs = _grads_struct(model, dev)
        b, keep = model._batch(*ctx.keep)
        wp, wn = _ws_ptr(ctx.ws)
        ps = model._params_struct()
        dl = None if dlogp is None else dlogp.contiguous().float()
        dc = None if dcat is None else dcat.contiguous().float()
        fn = nv.lib().xg_backward_ss if ctx.ss else nv.lib().xg_backward_xe
        run = _with_grad_event(model, ctx.run)
        nv.check(fn(_stream(), C.byref(d), C.byref(ps), C.byref(gs), C.byref(b), C.byref(run),
                    wp, wn, nv.ptr(dl), nv.ptr(dc)), "xg_backward_ss" if ctx.ss else "xg_backward_xe")
        model._pool.give(d, dev, ctx.ws)
        ctx.ws = None
        return (None,) * 9 + tuple(_grad_views(model, g))


class _XELossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, save, feats_rgb, feats_opfl, feat_mask, pos_feats, seq, seq_mask, cap_classes, class_mask,
                weight_class, *params):
        B, K, _ = feats_rgb.shape
        T = seq.shape[1]
        dev = feats_rgb.device
        d = model._dims(B, K, T)
        ws = model._pool.take(d, dev) if save else model._pool.shared(d, dev)
        wp, wn = _ws_ptr(ws)
        b, keep = model._batch(feats_rgb, feats_opfl, feat_mask, pos_feats, seq, seq_mask)
        cc = None if cap_classes is None else cap_classes.detach().contiguous().long()
        cm = None if class_mask is None else class_mask.detach().contiguous().float()
        losses = torch.empty(3, dtype=torch.float32, device=dev)
        ps, bn, run = model._params_struct(), model._bn_struct(), model._run(save)
        # (the counters' increment is independent of the forward: enqueued in FRONT of it, it is not one more small launch on the
        #  main stream between the loss and the start of the backward)
        model._bump_bn()
        try:
            nv.check(nv.lib().xg_xe_loss_fwd(_stream(), C.byref(d), C.byref(ps), C.byref(bn), C.byref(b), nv.ptr(cc),
                                             nv.ptr(cm), weight_class, C.byref(run), wp, wn, nv.ptr(losses)), "xg_xe_loss_fwd")
        except Exception:
            model._bump_bn(-1)                     # the call was refused: num_batches_tracked stays where it was
            raise
        ctx.model, ctx.d, ctx.ws, ctx.keep, ctx.run, ctx.saved_ws = model, d, ws, keep, run, save
        ctx.cc, ctx.cm, ctx.wc = cc, cm, weight_class
        model.last_losses = losses.detach()        # READ-ONLY for callers: it shares storage with the returned loss (no copy kernel)
        return losses[0]                           # (a view of the three-element result: no copy kernel)

    @staticmethod
    def backward(ctx, dloss):
        model, d = ctx.model, ctx.d
        dev = ctx.keep[0].device
        g, gs = _grads_struct(model, dev)
        b, keep = model._batch(*ctx.keep)
        wp, wn = _ws_ptr(ctx.ws)
        ps = model._params_struct()
        dl = dloss.detach().contiguous().float().to(dev)
        run = _with_grad_event(model, ctx.run)
        nv.check(nv.lib().xg_xe_loss_bwd(_stream(), C.byref(d), C.byref(ps), C.byref(gs), C.byref(b), nv.ptr(ctx.cc),
                                         nv.ptr(ctx.cm), ctx.wc, nv.ptr(dl), C.byref(run), wp, wn), "xg_xe_loss_bwd")
        model._pool.give(d, dev, ctx.ws)
        ctx.ws = None
        return (None,) * 11 + tuple(_grad_views(model, g))


class _RolloutFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, feats_rgb, feats_opfl, feat_mask, pos_feats, mode, uniforms, forced, temperature,
                need_grad, bn_update, *params):
        B, K, _ = feats_rgb.shape
        T = model.seq_length + 1
        dev = feats_rgb.device
        d = model._dims(B, K, T)
        ws = model._pool.take(d, dev) if need_grad else model._pool.shared(d, dev)
        wp, wn = _ws_ptr(ws)
        b, keep = model._batch(feats_rgb, feats_opfl, feat_mask, pos_feats)
        seq = torch.zeros(B, T - 1, dtype=torch.int64, device=dev)
        slp = torch.zeros(B, T - 1, dtype=torch.float32, device=dev)
        n = torch.zeros(1, dtype=torch.int32, device=dev)
        if mode == nv.XG_ROLLOUT_SAMPLE and uniforms is None:
            uniforms = torch.rand(T, B, device=dev, dtype=torch.float32)
        if uniforms is not None:
            uniforms = uniforms.detach().contiguous().float()
        if forced is not None:
            f = torch.zeros(B, T - 1, dtype=torch.int64, device=dev)
            f[:, :forced.shape[1]] = forced.to(dev)
            forced = f
        ps, bn, run = model._params_struct(), model._bn_struct(), model._run(need_grad)
        if not bn_update and model.training:   # batch statistics are used, the running buffers are left alone
            bn = nv.XgBnState()
        nv.check(nv.lib().xg_rollout(_stream(), C.byref(d), C.byref(ps), C.byref(bn), C.byref(b), C.byref(run), mode,
                                     nv.ptr(uniforms), nv.ptr(forced), temperature, wp, wn, nv.ptr(seq), nv.ptr(slp),
                                     nv.ptr(n)), "xg_rollout")
        if bn_update:
            model._bump_bn()
        ctx.model, ctx.d, ctx.ws, ctx.keep, ctx.run, ctx.need_grad = model, d, ws, keep, run, need_grad
        ctx.mark_non_differentiable(seq, n)
        if not need_grad:
            ctx.mark_non_differentiable(slp)
        return seq, slp, n

    @staticmethod
    def backward(ctx, dseq, dslp, dn):
        model, d = ctx.model, ctx.d
        if not ctx.need_grad:
            raise nv.XgError("rollout ran without saved activations")
        dev = ctx.keep[0].device
        g, gs = _grads_struct(model, dev)
        b, keep = model._batch(*ctx.keep)
        wp, wn = _ws_ptr(ctx.ws)
        ps = model._params_struct()
        T = d.T
        full = torch.zeros(d.B, T - 1, dtype=torch.float32, device=dev)
        full[:, :dslp.shape[1]] = dslp
        run = _with_grad_event(model, ctx.run)
        nv.check(nv.lib().xg_rollout_bwd(_stream(), C.byref(d), C.byref(ps), C.byref(gs), C.byref(b), C.byref(run),
                                         wp, wn, nv.ptr(full)), "xg_rollout_bwd")
        model._pool.give(d, dev, ctx.ws)
        ctx.ws = None
        return (None,) * 11 + tuple(_grad_views(model, g))


class _RolloutPairFunction(torch.autograd.Function):
    """Sampled rollout + greedy baseline of one SCST iteration as ONE batch of 2m rows (xg_rollout_pair); the sampled
    half's activations are compacted into an m-row workspace, so backward is xg_rollout_bwd of the sampled rollout alone."""

    @staticmethod
    def forward(ctx, model, feats_rgb, feats_opfl, feat_mask, pos_feats, uniforms, temperature, need_grad, *params):
        B, K, _ = feats_rgb.shape
        T = model.seq_length + 1
        dev = feats_rgb.device
        d1, d2 = model._dims(B, K, T), model._dims(2 * B, K, T)
        ws2 = model._pool.shared(d2, dev)
        wp2, wn2 = _ws_ptr(ws2)
        # the videos are handed over ONCE: the library runs the encoder on the m-row batch and repeats V / v2a(V) / the
        # initial state on the device (xg_rollout_pair_videos) -- no torch.cat of the inputs, half the encoder work
        fm = feat_mask
        if fm.dim() == 1:
            fm = fm.unsqueeze(0).expand(B, K)
        b1, keep1 = model._batch(feats_rgb, feats_opfl, fm, pos_feats)
        seq = torch.zeros(2 * B, T - 1, dtype=torch.int64, device=dev)
        slp = torch.zeros(2 * B, T - 1, dtype=torch.float32, device=dev)
        n = torch.zeros(2, dtype=torch.int32, device=dev)
        if uniforms is None:
            uniforms = torch.rand(T, B, device=dev, dtype=torch.float32)
        uniforms = uniforms.detach().contiguous().float()
        ps, bn, run = model._params_struct(), model._bn_struct(), model._run(need_grad)
        # rollout + compaction of the sampled half into the m-row workspace (what the backward reads) as one call: the encoder
        # runs there in the first place, and the sampled rows' logits are written there by the rollout itself
        ws1 = model._pool.take(d1, dev) if need_grad else model._pool.shared(d1, dev)
        wp1, wn1 = _ws_ptr(ws1)
        nv.check(nv.lib().xg_rollout_pair_videos(_stream(), C.byref(d2), C.byref(ps), C.byref(bn), C.byref(b1), C.byref(run),
                                                 nv.ptr(uniforms), temperature, wp2, wn2, C.byref(d1), wp1, wn1, 1 if need_grad else 0,
                                                 nv.ptr(seq), nv.ptr(slp), nv.ptr(n)), "xg_rollout_pair_videos")
        if not need_grad:
            ws1 = None
        model._bump_bn(2)                         # two sample() calls of the reference: two running-statistics updates (in the library)
        ctx.model, ctx.d, ctx.ws, ctx.run, ctx.need_grad = model, d1, ws1, run, need_grad
        ctx.keep = (feats_rgb, feats_opfl, feat_mask, pos_feats)
        gen, greedy = seq[:B], seq[B:]
        slp_s = slp[:B]
        ctx.mark_non_differentiable(gen, greedy, n)
        if not need_grad:
            ctx.mark_non_differentiable(slp_s)
        return gen, slp_s, greedy, n

    @staticmethod
    def backward(ctx, dgen, dslp, dgreedy, dn):
        model, d = ctx.model, ctx.d
        if not ctx.need_grad:
            raise nv.XgError("rollout ran without saved activations")
        dev = ctx.keep[0].device
        g, gs = _grads_struct(model, dev)
        b, keep = model._batch(*ctx.keep)
        wp, wn = _ws_ptr(ctx.ws)
        ps = model._params_struct()
        full = torch.zeros(d.B, d.T - 1, dtype=torch.float32, device=dev)
        full[:, :dslp.shape[1]] = dslp
        run = _with_grad_event(model, ctx.run)
        nv.check(nv.lib().xg_rollout_bwd(_stream(), C.byref(d), C.byref(ps), C.byref(gs), C.byref(b), C.byref(run),
                                         wp, wn, nv.ptr(full)), "xg_rollout_bwd")
        model._pool.give(d, dev, ctx.ws)
        ctx.ws = None
        return (None,) * 8 + tuple(_grad_views(model, g))


# ====================================================================== criteria
class _NLLFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logp, target, mask, mask2, roll):
        B, T, V = logp.shape
        logp = logp.contiguous()
        target = target.detach().contiguous().long()
        mask = mask.detach().contiguous().float()
        m2 = None if mask2 is None else mask2.detach().contiguous().float()
        sums = torch.empty(2, dtype=torch.float32, device=logp.device)
        nv.check(nv.lib().xg_nll_fwd(_stream(), nv.ptr(logp), nv.ptr(target), nv.ptr(mask), nv.ptr(m2), B, T, V, roll,
                                     nv.ptr(sums)), "xg_nll_fwd")
        ctx.meta = (B, T, V, roll, target, mask, m2, sums)
        return sums[0] / sums[1]

    @staticmethod
    def backward(ctx, dloss):
        B, T, V, roll, target, mask, m2, sums = ctx.meta
        dlogp = torch.empty(B, T, V, dtype=torch.float32, device=sums.device)
        dl = dloss.detach().reshape(1).contiguous().float().to(sums.device)      # device scalar: no sync, no extra pass
        nv.check(nv.lib().xg_nll_bwd(_stream(), nv.ptr(target), nv.ptr(mask), nv.ptr(m2), B, T, V, roll, nv.ptr(sums),
                                     1.0, nv.ptr(dl), nv.ptr(dlogp)), "xg_nll_bwd")
        return dlogp, None, None, None, None


class LanguageModelCriterion(nn.Module):
    """reference caption_src/SAModel.py:221-234: masked NLL with the target rolled left by one."""

    def forward(self, input, target, mask):
        return _NLLFunction.apply(input, target, mask, None, 1)


class ClassiferCriterion(nn.Module):
    """reference caption_src/SAModel.py:236-253: masked NLL, target not rolled, optional class mask."""

    def forward(self, input, target, mask, class_mask=None):
        return _NLLFunction.apply(input, target, mask, class_mask, 0)


class _RewardFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, slp, seq, reward, n):
        m, L = slp.shape
        slp_c = slp if slp.stride(1) == 1 else slp.contiguous()
        seq = seq.detach()
        seq = seq if (seq.stride(1) == 1 and seq.dtype == torch.int64) else seq.contiguous().long()
        reward = reward.detach()
        reward = reward.float() if reward.dtype != torch.float32 else reward
        if reward.dim() == 0:                  # one scalar for every element (broadcast like the reference's input * reward)
            reward = reward.reshape(1, 1)
        if reward.dim() == 1:
            # The reference flattens both tensors (SAModel.py:260-261: view(-1)) and never broadcasts, so a 1-D reward has no
            # reference meaning unless it is the full m*L vector.  Extensions: (m*L,) is the flattened (m, L) matrix; (1,) one
            # value for all; (m,) with m != L one value per video; (L,) with L != m one value per position.  m == L is ambiguous
            # and rejected -- pass (m, 1) or (1, L).
            k = reward.shape[0]
            if k == m * L and m > 1 and L > 1:
                reward = reward.reshape(m, L)
            elif k == 1:
                reward = reward.reshape(1, 1)
            elif m == L and k == m:
                raise nv.XgError("1-D reward of length m == L is ambiguous: pass (m, 1) per video or (1, L) per position")
            elif k == m:
                reward = reward.unsqueeze(1)
            elif k == L:
                reward = reward.unsqueeze(0)
            else:
                raise nv.XgError("1-D reward must have m, L or m*L elements")
        if reward.shape[0] not in (1, m):
            raise nv.XgError("reward must broadcast against the (m, L) log-probs")
        if reward.shape[1] == 1:
            rs_b, rs_t = (reward.stride(0) if reward.shape[0] == m else 0), 0
        else:
            if reward.shape[1] < L:
                raise nv.XgError("reward has fewer columns than the log-probs")
            rs_b, rs_t = (reward.stride(0) if reward.shape[0] == m else 0), reward.stride(1)
        nd = None if n is None else n.detach().reshape(-1)[:1].to(torch.int32)
        sums = torch.empty(2, dtype=torch.float32, device=slp.device)
        nv.check(nv.lib().xg_reward_fwd(_stream(), nv.ptr(slp_c), slp_c.stride(0), nv.ptr(seq), seq.stride(0), nv.ptr(reward),
                                        rs_b, rs_t, nv.ptr(nd), m, L, nv.ptr(sums)), "xg_reward_fwd")
        ctx.meta = (m, L, seq, reward, rs_b, rs_t, nd, sums)
        return sums[0] / sums[1]

    @staticmethod
    def backward(ctx, dloss):
        m, L, seq, reward, rs_b, rs_t, nd, sums = ctx.meta
        dslp = torch.empty(m, L, dtype=torch.float32, device=sums.device)
        dl = dloss.detach().reshape(1).contiguous().float().to(sums.device)
        nv.check(nv.lib().xg_reward_bwd(_stream(), nv.ptr(seq), seq.stride(0), nv.ptr(reward), rs_b, rs_t, nv.ptr(nd), m, L,
                                        nv.ptr(sums), nv.ptr(dl), nv.ptr(dslp), L), "xg_reward_bwd")
        return dslp, None, None, None


class RewardCriterion(nn.Module):
    """reference caption_src/SAModel.py:255-267: policy-gradient loss -sum(input * reward * mask) / sum(mask) with
    mask[:, 0] = 1, mask[:, t] = seq[:, t-1] > 0, as ONE HIP launch forward and one backward (xg_reward_fwd/bwd).
    ``n`` (optional device int tensor: the rollout's early-exit width, e.g. from ``model.sample(..., {'async': True})`` or
    ``scst_rollouts(..., trim=False)``) lets the caller pass full-width (m, L) rollout outputs without ever syncing on n;
    ``reward`` may be (m, n'), one value per video (m, 1), one per position (1, L), a scalar, or 1-D of length m / L when
    m != L (m == L is ambiguous and raises)."""

    def forward(self, input, seq, reward, n=None):
        reward = torch.as_tensor(reward, dtype=torch.float32, device=input.device)
        return _RewardFunction.apply(input, seq, reward, n)
