"""Training-side counterparts of the reference's driver utilities:
clip_gradient + Adam (caption_src/myutils.py:79-85, caption_src/starttrain.py:76,136-137) as ONE
fused HIP launch over the model's flat parameter buffer, and the data-parallel gradient
all-reduce (RCCL over xGMI; the reference itself is single-GPU)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _native as nv
from .model import _stream


_SHARED_STREAMS = {}


def shared_stream(kind):
    """ONE side stream per (device, purpose) for the whole process.  Every HIP stream is bound to a hardware queue, and the
    queues share the compute pipes: a fifth, sixth ... stream lands on the pipe of an earlier one, and a stream that sits on
    the main stream's pipe while it waits for an event stalls the main stream (measured: the second ClipAdam(overlap=True)
    of a process, whose fresh pool stream shared the main stream's pipe, made every iteration 2x slower: 6.1 -> 11.8 ms).
    So optimizers, gradient synchronisers and rollouts do not take a fresh stream from torch's pool each time they are
    constructed; they share these."""
    key = (torch.cuda.current_device(), kind)
    st = _SHARED_STREAMS.get(key)
    if st is None:
        st = _SHARED_STREAMS[key] = torch.cuda.Stream()
    return st


def dist_diagnosis():
    """One line that explains a first multi-GPU failure (RCCL init, a hung collective, a missing device): what this process sees.
    Printed by bench.py --gpus N and tests/test_gpu_dp2.py when the process group does not come up or the first collective fails."""
    import os
    env = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HIP_VISIBLE_DEVICES",
                                          "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_MAX_HW_QUEUES", "HSA_ENABLE_IPC_MODE_LEGACY",
                                          "NCCL_MAX_NCHANNELS", "NCCL_DEBUG")}
    try:
        rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:                       # pragma: no cover
        rccl = "unavailable (%r)" % (e,)
    try:
        ndev, cur = torch.cuda.device_count(), (torch.cuda.current_device() if torch.cuda.is_available() else None)
        names = sorted({torch.cuda.get_device_name(i) for i in range(ndev)})
    except Exception as e:                       # pragma: no cover
        ndev, cur, names = -1, None, [repr(e)]
    return "[xgate dist] rccl %s | torch %s hip %s | %d visible device(s) %s, current %s | %s" % (
        rccl, torch.__version__, getattr(torch.version, "hip", None), ndev, names, cur,
        " ".join("%s=%s" % (k, v) for k, v in env.items() if v is not None))


class ClipAdam:
    """optimizer = optim.Adam(model.parameters(), lr, weight_decay) + clip_gradient(optimizer, clip):
    elementwise clamp of every gradient to +-grad_clip, then Adam with torch defaults
    (betas 0.9/0.999, eps 1e-8; the reference ignores its --optim_* flags, starttrain.py:76).

    ``overlap=True``: the update of a parameter group starts as soon as its gradient is final instead of after the whole
    backward -- the update is HBM-bound, the backward's tail is matrix-core bound, so they overlap almost for free.  The
    flat buffers are in xg_param_name order [two_spatial_encoder.* | img_embed / lstmcore / embed | logit.* | classifer.*]
    and the library records when logit.* is final (XgRun.grad_event_head) and when everything but the encoder is
    (XgRun.grad_event): ``arm()`` before ``loss.backward()``, then ``step()`` issues three segment updates, two of them on
    a side stream behind those events.  Same arithmetic, element for element.  In a data-parallel run the segments are
    updated right behind their all-reduce (train.GradSync.finish).

    ``fused_zero=True``: the update leaves the gradient buffer at ZERO (xg_clip_adam_zero) instead of clamped, and the
    ``zero_grad()`` that opens the next iteration (starttrain.py:123) then has nothing to do -- it still clears the buffer
    if anything wrote to it in between (a backward without a step, an in-place edit of a ``.grad``).  Only the contents of
    ``.grad`` between ``step()`` and ``zero_grad()`` differ from the reference (clamped values there), which nothing reads."""

    def __init__(self, model, lr=4e-4, weight_decay=0.0, grad_clip=0.1, betas=(0.9, 0.999), eps=1e-8, overlap=False,
                 fused_zero=False, device_state=False):
        self.model, self.lr, self.wd, self.clip, self.betas, self.eps = model, lr, weight_decay, grad_clip, betas, eps
        flat = model.flat_parameters()
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.step_count = 0
        # device_state=True: the step counter, both bias corrections and the learning rate live in device memory
        # (xg_adam_tick / xg_clip_adam_dev), so that no launch argument changes from step to step: required for
        # GraphedXEStep (the iteration as a replayed HIP graph), same arithmetic otherwise
        self.device_state = bool(device_state)
        self._ticked = False
        if self.device_state:
            self._hyper = torch.zeros(4, dtype=torch.float32, device=flat.device)
            self._hyper[0] = lr
        self.overlap = bool(overlap)
        self.fused_zero = bool(fused_zero)
        self._zero_stamp = None                # (model._grad_writes, flat_grads._version) right after a zeroing update
        self._armed = False
        self._segments_done = False
        if self.overlap:
            self._split, self._head = _segment_bounds(model)
            self._event, self._event_head = torch.cuda.Event(), torch.cuda.Event()
            self._event.record(); self._event_head.record()      # torch creates the HIP events lazily
            self._side = shared_stream("update")
            model._overlap_optimizer = self

    def _grad_stamp(self):
        return (getattr(self.model, "_grad_writes", 0), self.model.flat_grads()._version)

    def zero_grad(self):
        if self.fused_zero and self._zero_stamp is not None and self._zero_stamp == self._grad_stamp():
            return                             # the last update left it zero and nothing has written to it since
        self.model.flat_grads().zero_()

    def set_lr(self, lr):                      # myutils.set_lr
        if self.device_state and lr != self.lr:
            self._hyper[0] = lr
        self.lr = lr

    def _tick(self):
        nv.check(nv.lib().xg_adam_tick(_stream(), nv.ptr(self._hyper), self.betas[0], self.betas[1]), "xg_adam_tick")
        self._ticked = True

    def arm(self):
        """overlap=True: call before loss.backward() (after a GradSync.arm(), if any: the events are shared)."""
        if self.device_state and not self._ticked:
            self._tick()                       # (on the main stream, ahead of the backward: every segment update runs behind it)
        if not self.overlap or torch.cuda.is_current_stream_capturing():      # (a captured iteration is single-stream)
            return
        if getattr(self.model, "_grad_event", None) is None:
            self.model._grad_event, self.model._grad_event_head = self._event, self._event_head
            self._own_events = True
        else:
            self._own_events = False
        self._armed = True

    def disarm(self):
        """Back to the single-stream update for THIS iteration: the segment updates of ``step()`` may only run behind the
        library's grad-ready events when nothing else touches the gradients between the backward and the update.  A plain
        (un-overlapped) all-reduce does -- it is enqueued after the backward, i.e. after those events -- so
        ``allreduce_gradients`` calls this and ``step()`` then updates everything in stream order behind the collective."""
        if self._armed and getattr(self, "_own_events", False):
            self.model._grad_event = None
            self.model._grad_event_head = None
        self._armed = False

    def update_segment(self, a, b):
        """clamp + Adam of flat[a:b] on the current stream (bias correction of the step in progress)."""
        if b <= a:
            return
        flat, g = self.model.flat_parameters(), self.model.flat_grads()
        if self.device_state:
            nv.check(nv.lib().xg_clip_adam_dev(_stream(), b - a, nv.ptr(flat[a:b]), nv.ptr(g[a:b]), nv.ptr(self.exp_avg[a:b]),
                                               nv.ptr(self.exp_avg_sq[a:b]), nv.ptr(self._hyper), self.betas[0], self.betas[1],
                                               self.eps, self.wd, self.clip, 1 if self.fused_zero else 0), "xg_clip_adam_dev")
            return
        fn = nv.lib().xg_clip_adam_zero if self.fused_zero else nv.lib().xg_clip_adam
        nv.check(fn(_stream(), b - a, nv.ptr(flat[a:b]), nv.ptr(g[a:b]), nv.ptr(self.exp_avg[a:b]),
                    nv.ptr(self.exp_avg_sq[a:b]), self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                    self.step_count, self.clip), "xg_clip_adam")

    def begin_step(self):
        self.step_count += 1
        if self.device_state:
            if not self._ticked:
                self._tick()
            self._ticked = False

    def step(self):
        n = self.model.flat_parameters().numel()
        early = False
        if self._segments_done:                # a data-parallel GradSync.finish updated every segment behind its all-reduce
            self._segments_done = False
        elif self.overlap and self._armed:
            self.begin_step()
            ev, ev_head = self.model._grad_event, self.model._grad_event_head
            main = torch.cuda.current_stream()
            a0, a1 = self._head
            self._side.wait_event(ev_head)     # the backward has been enqueued: these are the records it made
            with torch.cuda.stream(self._side):
                self.update_segment(a0, a1)
            self._side.wait_event(ev)
            with torch.cuda.stream(self._side):
                self.update_segment(self._split, a0)
                self.update_segment(a1, n)
                # every matrix the per-step products stream except the encoder's is final now: their packed tiles are
                # refreshed here, under the encoder's backward, instead of at the head of the next iteration
                if hasattr(self.model, "pack_early") and not _NO_EARLY_PACK:
                    self.model.pack_early()
                    early = True               # from here on this step writes the encoder's segment [0, split) only
            self.update_segment(0, self._split)          # the encoder's gradients: after the whole backward (stream order)
            main.wait_stream(self._side)
        else:
            self.begin_step()
            self.update_segment(0, n)
        if self._armed:
            if self._own_events:
                self.model._grad_event = None
                self.model._grad_event_head = None
            self._armed = False
        # the kernel wrote the flat buffer directly: re-pack the recurrent weights (all of them, unless pack_early covered the decoder's)
        self.model.mark_params_changed(only_encoder_since_pack_early=early)
        self._zero_stamp = self._grad_stamp() if self.fused_zero else None   # every segment has been updated (and zeroed)

    def state_dict(self):
        return dict(exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, step=self.step_count, lr=self.lr)

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count, self.lr = sd["step"], sd["lr"]
        if self.device_state:
            self._hyper[0] = self.lr
            self._hyper.view(torch.int32)[3] = int(self.step_count)


class GraphedXEStep:
    """One teacher-forced XE training iteration -- zero_grad, fused forward + loss (SAModel.xe_loss), backward, clamp + Adam
    and the re-pack of the recurrent weights (starttrain.py:123-137) -- captured ONCE as a HIP graph and replayed: about
    350 kernel launches become one graph launch (host: 0.09 ms per iteration instead of 2.4 ms).  All shapes are static and
    nothing in the iteration syncs with the host, which is what makes it capturable; the step-dependent Adam scalars live in
    device memory (ClipAdam(device_state=True)).

    The captured iteration is SINGLE-STREAM: on ROCm 7.0 a capture that forks onto the library's side streams either crashes
    in hipStreamEndCapture (backward) or replays node by node (forward: 5.5 ms and 2.7 ms of host time against 2.4 ms eager),
    so the capture runs with XgRun.aux = NULL and the stream-ordered update.  That gives up the three-stream overlap: 7.05 ms
    per iteration against 6.10 ms for the eager loop on an idle host (configs[1], tools/graph_probe.py).  Use it when the
    host cannot keep up with ~350 launches per iteration (many ranks per host, a busy CPU); otherwise the eager loop is
    faster.  bench.py times the eager loop (--graph selects this one).

    ``batch``: dict of CUDA tensors (feats_rgb, feats_opfl, feat_mask, pos_feats, seq, seq_mask [, cap_classes, class_mask]);
    they are the graph's STATIC inputs: write new data into them with ``copy_`` (``load(batch)``) and call the object.
    Restrictions: drop_prob_lm == 0 (the dropout seed is a launch argument), ss_prob == 0, single process (the RCCL
    all-reduce is not captured).  Returns the loss tensor (static: overwritten by each replay)."""

    def __init__(self, model, optimizer, batch, weight_class=0.0, warmup=2):
        if not optimizer.device_state:
            raise ValueError("GraphedXEStep needs ClipAdam(device_state=True)")
        if model.drop_prob_lm > 0 and model.training:
            raise ValueError("GraphedXEStep: drop_prob_lm must be 0 (the dropout seed is a launch argument)")
        self.model, self.opt, self.x, self.wc = model, optimizer, dict(batch), float(weight_class)
        self.graph, self.loss = None, None
        self._capture(warmup)

    def _iteration(self):
        m, o, x = self.model, self.opt, self.x
        o.zero_grad()
        loss = m.xe_loss(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], x["seq"], x["seq_mask"],
                         x.get("cap_classes"), x.get("class_mask"), self.wc)
        o.arm()
        loss.backward()
        o.step()
        m._packed_ptr()                        # the re-pack belongs to the iteration (it would otherwise open the next one)
        return loss

    def _capture(self, warmup):
        m, o = self.model, self.opt
        # the warm-up iterations (workspaces, side-stream handles, LDS opt-ins must exist before the capture) really train:
        # snapshot everything they touch and put it back, so that building the graph leaves model and optimizer untouched
        bufs = [b for b in m.buffers()]
        snap = [t.clone() for t in (m.flat_parameters(), o.exp_avg, o.exp_avg_sq, o._hyper, m.flat_grads())] + [b.clone() for b in bufs]
        step0 = o.step_count
        side = shared_stream("graph")
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._iteration()
            side.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=side):
                self.loss = self._iteration()      # (recorded, not executed)
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad():
            for t, c in zip([m.flat_parameters(), o.exp_avg, o.exp_avg_sq, o._hyper, m.flat_grads()] + bufs, snap):
                t.copy_(c)
        o.step_count = step0
        m.mark_params_changed()
        m._packed_ptr()                            # the shadow of the restored weights (later re-packs are inside the graph)

    def load(self, batch):
        for k, v in batch.items():
            if k in self.x and torch.is_tensor(v):
                self.x[k].copy_(v)

    def __call__(self):
        self.graph.replay()
        self.opt.step_count += 1               # (host mirror of the device-side step counter)
        return self.loss


import os as _os
_FORCE = _os.environ.get("XG_FORCE_DIST") == "1"      # run the collective even at world size 1 (single-GPU smoke of the path)


_NO_EARLY_PACK = False      # (tests flip it: the early-repack negative control)
_SKIP_COLLECTIVE = False    # measurement switch (bench.py: exposed communication = iteration with - without the collective)


def _reduce(t, world, group):
    """sum over ranks / world, in place.  RCCL averages inside the collective; gloo (CPU tests) has no AVG."""
    import torch.distributed as dist
    if _SKIP_COLLECTIVE:
        return
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t.mul_(1.0 / world)


def allreduce_gradients(model, group=None):
    """Data parallel by video: all-reduce(sum) / world of the flat gradient buffer BEFORE the clamp.
    No other collective; BatchNorm statistics stay per replica.  ONE collective -- or, when a GradSync armed this
    backward, two: the part that was final before the CG encoder's backward has then already been started under it."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return
    world = dist.get_world_size(group)
    if world == 1 and not _FORCE:
        return
    sync = getattr(model, "_grad_sync", None)
    if sync is not None and sync.armed:
        sync.finish(group)
        return
    # plain path: ONE collective after the backward.  An armed overlapping optimizer would run its segment updates behind
    # events recorded DURING the backward, i.e. on un-averaged gradients (and, with fused_zero, leave zeros for this
    # collective to average): it must take the stream-ordered update instead.
    opt = getattr(model, "_overlap_optimizer", None)
    if opt is not None:
        opt.disarm()
    _reduce(model.flat_grads(), world, group)


def _segment_bounds(model):
    """(first non-encoder element, [logit.weight, end of logit.bias incl. padding)) of the flat buffers."""
    model._ensure_flat()
    first_other = next(n for n in nv.PARAM_NAMES if not n.startswith("two_spatial_encoder."))
    split = model._slices[first_other][0]
    assert all(n.startswith("two_spatial_encoder.") == (model._slices[n][0] < split) for n in nv.PARAM_NAMES)
    lw, lb = model._slices["logit.weight"], model._slices["logit.bias"]
    assert lb[0] > lw[0]
    return split, (lw[0], lb[0] + (lb[1] + 63) // 64 * 64)


def bucket_plan(split, head, numel):
    """The collectives of one GradSync iteration, in issue order, as (first, last, event) element ranges of the flat gradient
    buffer: event "head" = behind XgRun.grad_event_head, "rest" = behind XgRun.grad_event, "end" = behind the whole backward.
    A pure function of the buffer layout, so every rank issues the same collectives in the same order with the same bounds
    (tests/test_dp_gloo.py checks the partition and the order at world size 8)."""
    a0, a1 = head
    plan = [(a0, a1, "head"), (split, a0, "rest")]
    if a1 < numel:
        plan.append((a1, numel, "rest"))
    plan.append((0, split, "end"))
    return plan


class GradSync:
    """Overlaps the gradient all-reduce with the backward pass.  The flat gradient buffer is in xg_param_name order:
    [two_spatial_encoder.* | img_embed / lstmcore / embed | logit.* | classifer.*], and the backward pass finishes it
    roughly in that order reversed.  The library records two events (XgRun.grad_event_head / grad_event):
      * when logit.weight / logit.bias gradients are final (a third of the bytes; before the reverse-time decoder loop),
      * when everything except the CG encoder's gradients is final (the encoder's recurrent backward is what remains),
    and the all-reduce of each part is issued on a side stream that waits for its event, so RCCL's rings run under the
    remaining backward; the encoder prefix follows after the backward.  Same sums, same 1/world; four collectives in a
    fixed order on every rank (logit | middle | classifer | encoder).  Usage: sync = GradSync(model); per iteration:
    sync.arm(); loss.backward(); allreduce_gradients(model); optimizer.step().
    Expected exposure at 8 GPUs (144.5 MB, ring all-reduce, ~300 GB/s bus bandwidth): 0.24 ms (logit) + 0.45 ms (middle) run
    hidden under ~3 ms / ~1.3 ms of remaining backward; the 26 MB encoder part (~0.15 ms) is what stays exposed."""

    def __init__(self, model):
        self.model = model
        self.split, self.head = _segment_bounds(model)
        self.event, self.event_head = torch.cuda.Event(), torch.cuda.Event()
        self.event.record(); self.event_head.record()   # torch creates the HIP events lazily: make the handles exist
        self.side = shared_stream("allreduce")
        self.armed = False
        model._grad_sync = self

    def arm(self):
        """Call before loss.backward(): the next backward records the events."""
        self.model._grad_event = self.event
        self.model._grad_event_head = self.event_head
        self.armed = True

    def finish(self, group=None):
        import torch.distributed as dist
        world = dist.get_world_size(group)
        g = self.model.flat_grads()
        main = torch.cuda.current_stream()
        # an overlapping optimizer (ClipAdam(overlap=True), armed) updates each segment right behind its all-reduce
        opt = getattr(self.model, "_overlap_optimizer", None)
        opt = opt if (opt is not None and opt._armed) else None
        if opt is not None:
            opt.begin_step()
        upd = (lambda a, b: opt.update_segment(a, b)) if opt is not None else (lambda a, b: None)
        waited = None
        for lo, hi, ev in bucket_plan(self.split, self.head, g.numel()):
            if ev == "end":                          # after the whole backward (main stream order)
                _reduce(g[lo:hi], world, group)
                upd(lo, hi)
                continue
            if ev != waited:                         # the backward has been enqueued: these are the records it made
                self.side.wait_event(self.event_head if ev == "head" else self.event)
                waited = ev
            with torch.cuda.stream(self.side):
                _reduce(g[lo:hi], world, group)
                upd(lo, hi)
        main.wait_stream(self.side)
        if opt is not None:
            opt._segments_done = True
        self.model._grad_event = None
        self.model._grad_event_head = None
        self.armed = False


def broadcast_parameters(model, src=0, group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or _FORCE):
        dist.broadcast(model.flat_parameters(), src=src, group=group)
        for b in model.buffers():
            dist.broadcast(b, src=src, group=group)


def shard_batch(x: dict, rank: int, world: int):
    """rank r takes videos r::world."""
    return {k: v[rank::world].contiguous() for k, v in x.items()}
