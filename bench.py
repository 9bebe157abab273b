#!/usr/bin/env python3
"""bench.py -- decoder timesteps/sec (train fwd+bwd) at batch 128, MSRVTT-shaped 26 x (1536+1024).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full teacher-forced XE training iteration of BASELINE.json configs[1] on each rank's
batch of 128 videos: zero_grad -> CG encoder fwd -> 21 decoder steps -> both heads + loss -> full
backward -> (N>1: ONE RCCL all-reduce of the flat gradient) -> clip + Adam.  Inputs are synthetic
 and already resident in HBM.  value = N * B * T * K / max-over-ranks wall time.

Besides the contract line it reports
  roofline     : the decoder-step launch group (xg_step_fwd = attention + POS gate + 2 LSTM cells, the
                 unit of SURVEY.md 8d) timed with stream events; ALGORITHMIC bytes / duration vs 8 TB/s
  cpu_baseline : the oracle (CPU restatement of the reference, v2a recomputed per step like the
                 reference) timed on this box's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # launcher duty, before the HIP runtime starts (see controllable_xgating_amd/__init__.py)
import torch  # noqa: E402


def synth_inputs(B, K, L, V, R, F1, F2, C_, seed, device):
    """feats U[0,1), pos U(-1,1), seq[:,0]=0, words UniformInt[2,V), masks of ones."""
    g = torch.Generator().manual_seed(seed)
    x = dict(
        feats_rgb=torch.rand(B, K, F1, generator=g),
        feats_opfl=torch.rand(B, K, F2, generator=g),
        feat_mask=torch.ones(B, K),
        pos_feats=torch.rand(B, R, generator=g) * 2 - 1,
        seq=torch.cat([torch.zeros(B, 1, dtype=torch.int64), torch.randint(2, V, (B, L), generator=g)], 1),
        seq_mask=torch.ones(B, L + 1),
        cap_classes=torch.randint(0, C_, (B, L + 1), generator=g),
        class_mask=torch.ones(B, L + 1),
    )
    return {k: v.to(device) for k, v in x.items()}


def step_bytes(B, K, R, A, E, save, elem=4):
    """ALGORITHMIC bytes of one decoder step: lstmcore weights once + per-video streams.  elem = 2 for the
    bf16 configuration (SURVEY.md 8d prices C5 at 2-byte elements: 84.93 MB per step)."""
    w_core = (2 * R * A + A) + (A + 1) + (E * R + R) + ((E + 2 * R) * 4 * R + 12 * R) + (3 * R * 4 * R + 12 * R)
    per = K * A + K * R + 4 * R + 4 * R + E + R + 1
    s_save = (2 * 4 * R + K + R + R + 2 * R) if save else 0
    return elem * (w_core + B * (per + s_save))


def step_flops(B, K, R, A, E):
    """fp32 multiply-adds x2 of one decoder step: h2a, POS gate, the two cells' six products, attention scores + context."""
    return 2.0 * B * (2 * R * A + E * R + (E + 2 * R) * 4 * R + 3 * R * 4 * R + K * A + K * R)


def iteration_flops(cfg):
    """Multiply-adds x 2 of ONE teacher-forced training iteration (forward; the backward is counted as twice that: a data-gradient
    and a weight-gradient product for every forward product): encoder input side and recurrence, cross gates + fusion, v2a(V), the
    token side, T decoder steps, vocabulary and category heads.  Elementwise work, attention tanh / softmax and the update are not
    counted -- this prices the iteration against the matrix roof only."""
    B, K, R, A, E, V, C, T = (cfg[k] for k in ("B", "K", "R", "A", "E", "V", "C", "L"))
    T = T + 1
    F1, F2, N = cfg["F1"], cfg["F2"], cfg["B"] * cfg["K"]
    enc = 2.0 * N * R * (F1 + F2) + 2 * 2.0 * N * 4 * R * R + K * 2 * 2.0 * B * 4 * R * R + 2 * 2.0 * N * R * R + 2.0 * N * R * 2 * R
    dec = 2.0 * N * A * R + T * step_flops(B, K, R, A, E)
    heads = 2.0 * T * B * R * V + 2.0 * T * B * R * C
    return 3.0 * (enc + dec + heads)


def measure_step_group(model, x, reps=200):
    """Average duration of ONE decoder-step launch group (xg_step_fwd) with events on the library's stream."""
    from controllable_xgating_amd import _native as nv
    from controllable_xgating_amd.model import _stream, _ws_ptr
    B, K = x["feats_rgb"].shape[:2]
    with torch.no_grad():
        V = model.encode(x["feats_rgb"], x["feats_opfl"], x["feat_mask"])
        st = model.init_hidden(V, x["feat_mask"])
        state = torch.cat([st[0][0], st[0][1], st[1][0], st[1][1]], 0).contiguous()
        d = model._dims(B, K, 1)
        ps, run = model._params_struct(), model._run(False)
        vproj = torch.empty(B, K, model.att_size, device=V.device)
        nv.check(nv.lib().xg_vproj(_stream(), C.byref(d), C.byref(ps), nv.ptr(V), nv.ptr(vproj), C.byref(run)), "xg_vproj")
        ws = model._pool.shared(d, V.device)
        wp, wn = _ws_ptr(ws)
        tok = x["seq"][:, 1].contiguous()
        pos = x["pos_feats"].contiguous()

        def call():
            nv.check(nv.lib().xg_step_fwd(_stream(), C.byref(d), C.byref(ps), nv.ptr(tok), None, nv.ptr(V), nv.ptr(vproj),
                                          nv.ptr(pos), C.byref(run), 0, wp, wn, nv.ptr(state), None, None), "xg_step_fwd")
        for _ in range(10):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps


def measure_traffic(timeout_s=150):
    """HBM bytes per decoder-step launch group measured NOW: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE -- separate
    passes, no tracing, as MI355X_MICROARCH.md's HBM section prescribes) over tools/step_group_run.py, FETCH_SIZE doubled
    (gfx950 counts a wide coalesced read at half its bytes), WRITE_SIZE as reported.  Returns (bytes, source) or (None, why)."""
    import shutil, subprocess, tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import pmc_traffic
        tmp = tempfile.mkdtemp(prefix="xg_pmc_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        dirs = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            dirs[counter] = os.path.join(tmp, counter)
            r = subprocess.run(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", dirs[counter], "--",
                                sys.executable, os.path.join(ROOT, "tools", "step_group_run.py"), "40"],
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s failed: %s" % (counter, (r.stderr or r.stdout)[-300:])
        fetch, n1 = pmc_traffic.per_kernel(dirs["FETCH_SIZE"], "FETCH_SIZE")
        write, _ = pmc_traffic.per_kernel(dirs["WRITE_SIZE"], "WRITE_SIZE")
        tot = 0.0
        for k, mult in pmc_traffic.STEP.items():
            if k not in fetch or k not in write:
                return None, "no %s dispatches in the PMC pass" % k
            tot += (fetch[k] * 2.0 + write[k]) * 1024 * mult
        shutil.rmtree(tmp, ignore_errors=True)
        return int(round(tot)), "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run (FETCH x2 gfx950 correction; %d dispatches)" % sum(n1.values())
    except Exception as e:                                   # never lose the bench line over the counters
        return None, "PMC passes failed: %r" % (e,)


def load_traffic():
    """HBM bytes per decoder-step launch group from the committed PMC passes (tools/pmc_traffic.py)."""
    try:
        files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_step_traffic.json"))
        with open(os.path.join(ROOT, "profiles", files[-1])) as f:
            return json.load(f)["hbm_bytes_per_step"], "profiles/" + files[-1]
    except Exception:
        return None, None


def cpu_baseline(cfg, budget_s=20.0, hip_model=None, hip_x=None, p=0.0, seed=0):
    """Oracle (port of the reference's CPU path; reference-faithful: v2a(V) recomputed every step).
    Timed at two thread counts (all host cores, and 8 like SURVEY.md's anchors); the faster one is `value`.
    With `hip_model` the leg is also the run's parity check: the HIP model takes the oracle's procedural weights and its
    XE loss on the very same batch is compared with the oracle's (`parity` in the result).
    budget_s <= 0 (the secondary lines): ONE forward + backward at 8 threads, no warm-up -- the parity check plus a
    one-iteration rate.
    p > 0 (train-mode dropout, the reference's default 0.5: myopts.py:37): the oracle draws its masks from the integer hash of
    (seed, site, step, element) and the HIP model is pinned to the same seed (SAModel.dropout_seed), so both see the very same
    masks (the machinery of tests/test_gpu_parity.py:test_dropout_masks_match_oracle_hash)."""
    from oracle import paramgen as pg
    d = pg.make_dims(B=cfg["B"], K=cfg["K"], R=cfg["R"], A=cfg["A"], E=cfg["E"], V=cfg["V"], C=cfg["C"], L=cfg["L"],
                     F1=cfg["F1"], F2=cfg["F2"])
    Pn = pg.make_params(d)
    ncores = torch.get_num_threads()
    res = []
    for nt in (sorted({min(8, ncores), ncores}) if budget_s > 0 else [min(8, ncores)]):
        torch.set_num_threads(nt)
        res.append(_cpu_baseline_once(cfg, d, Pn, budget_s / 2, p, seed))
    torch.set_num_threads(ncores)
    best = max(res, key=lambda r: r["value"])
    best["sample"] += "; all thread counts tried: " + ", ".join("%d threads -> %.1f/s" % (r["cores"], r["value"]) for r in res)
    oracle_loss = best.pop("loss")
    for r in res:
        r.pop("loss", None)
    if hip_model is not None:
        hip_model.load_state_dict({k: torch.from_numpy(v) for k, v in Pn.items()}, strict=False)
        hip_model.train()
        hip_model.dropout_seed = seed if p > 0.0 else None
        with torch.no_grad():
            hl = hip_model.xe_loss(hip_x["feats_rgb"], hip_x["feats_opfl"], hip_x["feat_mask"], hip_x["pos_feats"], hip_x["seq"],
                                   hip_x["seq_mask"])
        hip_model.dropout_seed = None
        best["parity"] = {"oracle_loss": oracle_loss, "hip_loss": float(hl.item()), "delta": abs(float(hl.item()) - oracle_loss)}
    return best


def _cpu_baseline_once(cfg, d, Pn, budget_s, p=0.0, seed=0):
    from oracle import xgate_oracle as xo
    P = xo.to_torch_params(Pn, requires_grad=True)
    x = {k: v.cpu() for k, v in synth_inputs(cfg["B"], cfg["K"], cfg["L"], cfg["V"], cfg["R"], cfg["F1"], cfg["F2"], cfg["C"],
                                           0, "cpu").items()}
    running = xo.new_running(d)

    def it():
        for p_ in P.values():
            p_.grad = None
        logp, cat, _ = xo.forward_xe(P, x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], x["seq"],
                                     x["seq_mask"], train=True, running=running, hoist=False, p=p, seed=seed)
        loss = xo.lm_criterion(logp, x["seq"], x["seq_mask"])
        loss.backward()
        return float(loss.item())
    t0 = time.time(); loss0 = it(); warm = time.time() - t0
    T = cfg["L"] + 1
    if budget_s <= 0:                            # the one iteration is the sample
        return dict(value=round(cfg["B"] * T / warm, 1), unit="decoder timesteps/s", cores=torch.get_num_threads(), kind="port",
                    sample="1 full XE fwd+bwd iteration of the same workload (B=%d, K=%d, R=%d, T=%d, V=%d), no warm-up; oracle with "
                           "v2a(V) recomputed per step like the reference" % (cfg["B"], cfg["K"], cfg["R"], T, cfg["V"]),
                    ms_per_step=round(warm * 1e3, 1), loss=loss0)
    n, t0 = 0, time.time()
    while True:
        it(); n += 1
        if time.time() - t0 > budget_s or n >= 4:
            break
    dt = (time.time() - t0) / n
    return dict(value=round(cfg["B"] * T / dt, 1), unit="decoder timesteps/s", cores=torch.get_num_threads(), kind="port",
                sample="%d full XE fwd+bwd iterations of the same workload (B=%d, T=%d, V=%d) after 1 warm-up (%.1f s); "
                       "oracle with v2a(V) recomputed per step like the reference" % (n, cfg["B"], T, cfg["V"], warm),
                ms_per_step=round(dt * 1e3, 1), loss=loss0)


def pin_host_thread(local_rank, world):
    """One training process enqueues ~350 launches per iteration (2 ms of host time); eight of them on one host must not share
    cores.  Pin this rank's process to its own slice of the cores it may use (os.sched_setaffinity by LOCAL_RANK), unless the
    launcher already narrowed the affinity (then keep it).  Returns the core list in use (reported per rank in comm.per_rank)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
        if os.environ.get("XG_NO_PIN") is not None or len(allowed) < 2 * max(world, 1):
            return allowed
        per = max(2, len(allowed) // max(world, 1))
        mine = allowed[local_rank * per:(local_rank + 1) * per] or allowed
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(len(mine), 8)))
        return mine
    except Exception:                            # (not Linux / not permitted: run unpinned)
        return None


def core_ranges(cores):
    """[0, 1, 2, 3, 8, 9] -> '0-3,8-9' (None stays None)."""
    if not cores:
        return None
    cores = sorted(cores)
    out, a, b = [], cores[0], cores[0]
    for c in cores[1:]:
        if c == b + 1:
            b = c
            continue
        out.append("%d-%d" % (a, b) if b > a else str(a))
        a = b = c
    out.append("%d-%d" % (a, b) if b > a else str(a))
    return ",".join(out)


def cap_rccl_channels():
    """RCCL's collectives run as kernels that need CUs; during the backward every CU also hosts one workgroup of a persistent
    background product and the latency-bound chain kernels.  Whether fewer channels (= fewer RCCL workgroups) help or starve the
    ring over xGMI (7 links x ~153 GB/s per GPU) cannot be decided without an N > 1 run, and none has happened: RCCL's own choice
    stands unless the launcher sets NCCL_MAX_NCHANNELS or XG_RCCL_CAP=<n> (which this sets before the communicator comes up).
    Either way the setting is reported with the scaling line (comm.rccl_channel_cap) next to the channel count RCCL logs."""
    if "NCCL_MAX_NCHANNELS" in os.environ:
        return {"NCCL_MAX_NCHANNELS": os.environ["NCCL_MAX_NCHANNELS"], "set_by": "environment"}
    cap = os.environ.get("XG_RCCL_CAP")
    if cap:
        os.environ["NCCL_MAX_NCHANNELS"] = str(int(cap))
        return {"NCCL_MAX_NCHANNELS": str(int(cap)), "set_by": "XG_RCCL_CAP"}
    return {"NCCL_MAX_NCHANNELS": None, "set_by": "RCCL default (XG_RCCL_CAP=<n> to cap)"}


def rccl_debug_setup(rank):
    """Ask RCCL for its INIT log in a private file (unless the user already directs NCCL_DEBUG somewhere) so that the channel
    count of the communicator can be reported with the scaling line."""
    if "NCCL_DEBUG" in os.environ:
        return None
    path = "/tmp/xg_rccl_%d_%d.log" % (os.getpid(), rank)
    os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT", NCCL_DEBUG_FILE=path)
    return path


def rccl_channels(path):
    """{"coll": c, "p2p": p, ...} parsed from RCCL's INIT log ('N coll channels, ... M p2p channels'), or None."""
    import re
    try:
        txt = open(path, errors="replace").read()
    except Exception:
        return None
    env = {k: os.environ[k] for k in ("NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS", "NCCL_NCHANNELS_PER_PEER") if k in os.environ}
    m = re.findall(r"(\d+) coll channels, (\d+) collnet channels, (\d+) nvls channels, (\d+) p2p channels", txt)
    if m:
        c = m[-1]
        return {"coll": int(c[0]), "collnet": int(c[1]), "nvls": int(c[2]), "p2p": int(c[3]), "env": env}
    m = re.findall(r"Channel (\d+)/(\d+)", txt)          # older log format: one 'Channel ii/nn : ring' line per channel
    if m:
        return {"coll": max(int(b) for _, b in m), "env": env}
    # nothing recognised: hand back the lines that mention channels so the first scaling run can still be read
    lines = [ln.strip()[-160:] for ln in txt.splitlines() if "hannel" in ln][:6]
    return {"coll": None, "env": env, "log_lines": lines}


def scst_parity(model, cfg, x, reward_b):
    """SCST leg of the self-check: the HIP model takes the oracle's procedural weights, runs the paired rollout, and the oracle
    REPLAYS the tokens the HIP sampler drew (oracle/xgate_oracle.py:sample, mode='replay'); the RewardCriterion losses must
    agree.  (Only the checker runs the oracle; about 5 s of CPU.)"""
    from oracle import paramgen as pg
    from oracle import xgate_oracle as xo
    from controllable_xgating_amd import RewardCriterion
    from controllable_xgating_amd.driver import scst_rollouts
    d = pg.make_dims(B=cfg["B"], K=cfg["K"], R=cfg["R"], A=cfg["A"], E=cfg["E"], V=cfg["V"], C=cfg["C"], L=cfg["L"],
                     F1=cfg["F1"], F2=cfg["F2"])
    Pn = pg.make_params(d, logit_gain=1.0)
    Pn["logit.bias"] = Pn["logit.bias"].copy()
    Pn["logit.bias"][0] += 7.0                              # EOS mass 5-10 % per step: rows finish anywhere in the rollout
    model.load_state_dict({k: torch.from_numpy(v) for k, v in Pn.items()}, strict=False)
    model.train()
    with torch.no_grad():
        gen, slp, greedy, n = scst_rollouts(model, x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], trim=False)
        loss_h = float(RewardCriterion()(slp, gen, reward_b, n=n[:1]).item())
    n_s = int(n[0].item())
    forced = gen[:, :n_s].cpu()
    xc = {k: v.cpu() for k, v in x.items()}
    with torch.no_grad():
        s_o, lp_o = xo.sample(xo.to_torch_params(Pn), xc["feats_rgb"], xc["feats_opfl"], xc["feat_mask"], xc["pos_feats"], cfg["L"],
                              mode="replay", forced=forced, train=True, running=xo.new_running(d))
        loss_o = float(xo.reward_criterion(lp_o, s_o, reward_b.cpu().expand(-1, s_o.shape[1])).item())
    return {"oracle_loss": loss_o, "hip_loss": loss_h, "delta": abs(loss_h - loss_o), "rollout_width": n_s}


def run_workload(args, workload, precision, steps, warmup, ctx, cpu_leg=True, pmc=True, comm_diag=True, drop=None):
    """One timed workload on this rank; returns (json-able dict | None on ranks != 0, parity failure text | None)."""
    world, rank, dev, use_dist = ctx["world"], ctx["rank"], ctx["dev"], ctx["use_dist"]
    drop = args.drop if drop is None else drop
    if use_dist:
        import torch.distributed as dist
    from controllable_xgating_amd import LanguageModelCriterion, RewardCriterion, SAModel, make_opt
    from controllable_xgating_amd import train as tr
    from controllable_xgating_amd.train import ClipAdam, GradSync, allreduce_gradients, broadcast_parameters
    from controllable_xgating_amd.driver import scst_rollouts

    cfg = dict(B=args.batch, K=26, R=512, A=1536, E=468, V=20000, C=14, L=20, F1=1536, F2=1024)
    if workload == "scst":
        cfg.update(B=64 if args.batch == 128 else args.batch, L=30)
    if workload == "xe5":          # BASELINE.json configs[4]: hidden 1024, 40 frames, vocab 20k
        cfg.update(K=40, R=1024)
    T = cfg["L"] + 1
    opt = make_opt(None, vocab_size=cfg["V"], seq_length=cfg["L"], drop_prob_lm=drop, precision=precision,
                   rnn_size=cfg["R"], att_size=cfg["A"], input_encoding_size=cfg["E"], feat_size=cfg["F1"], feat_size2=cfg["F2"])
    model = SAModel(opt).to(dev)
    model.train()
    broadcast_parameters(model)
    x = synth_inputs(cfg["B"], cfg["K"], cfg["L"], cfg["V"], cfg["R"], cfg["F1"], cfg["F2"], cfg["C"], seed=rank, device=dev)
    # clamp + Adam start per parameter group as soon as its gradient is final (XG_NO_UPDATE_OVERLAP=1: after the backward)
    # --graph: the fixed-shape XE iteration replayed as ONE HIP graph (train.GraphedXEStep).  Not the default: the capture has to
    # be single-stream on this ROCm, which costs more GPU time (7.05 vs 6.10 ms) than the 2.4 ms of host work it removes
    use_graph = args.graph and not use_dist and workload in ("xe", "xe5") and args.path == "fused" and drop == 0.0
    optim = ClipAdam(model, lr=4e-4, grad_clip=0.1, overlap=os.environ.get("XG_NO_UPDATE_OVERLAP") is None,
                     fused_zero=True, device_state=use_graph)
    # data parallel: most of the gradient all-reduce runs under the CG encoder's backward (XG_NO_GRAD_OVERLAP=1: one
    # all-reduce after the backward)
    sync = None
    if use_dist and os.environ.get("XG_NO_GRAD_OVERLAP") is None:
        try:
            sync = GradSync(model)
        except Exception as e:                   # never lose the run over the overlap: fall back to one all-reduce
            print("GradSync unavailable (%s): plain all-reduce" % e, file=sys.stderr)
    state = {"overlap_comm": True}              # data parallel: bucketed all-reduce under the backward (GradSync) vs one collective after it
    crit = LanguageModelCriterion()
    rl_crit = RewardCriterion()
    reward_b = torch.randn(cfg["B"], 1, generator=torch.Generator().manual_seed(7)).to(dev)   # CIDEr stubbed (configs[2])

    def step_scst():
        optim.zero_grad()
        gen, slp, greedy, n = scst_rollouts(model, x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], trim=False)     # no host sync anywhere in the iteration
        loss = rl_crit(slp, gen, reward_b, n=n[:1])
        if sync is not None and state["overlap_comm"]:
            sync.arm()
        optim.arm()
        loss.backward()
        allreduce_gradients(model)
        optim.step()
        return loss

    def step():
        if workload == "scst":
            return step_scst()
        optim.zero_grad()
        if args.path == "fused":
            loss = model.xe_loss(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], x["seq"], x["seq_mask"])
        else:
            logp, _ = model(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], x["seq"], x["seq_mask"])
            loss = crit(logp, x["seq"], x["seq_mask"])
        if sync is not None and state["overlap_comm"]:
            sync.arm()
        optim.arm()
        loss.backward()
        allreduce_gradients(model)
        optim.step()
        return loss

    eager_step = step
    if use_graph:
        try:
            from controllable_xgating_amd.train import GraphedXEStep
            gstep = GraphedXEStep(model, optim, x)
            step = lambda: gstep()               # noqa: E731
        except Exception as e:                   # never lose the run over the capture
            print("HIP graph capture unavailable (%r): eager launches" % (e,), file=sys.stderr)
            use_graph = False

    def timed(n):
        """n iterations bracketed by barrier + synchronize on both sides; (this rank's seconds, host enqueue seconds, last loss)"""
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            loss = step()
        t_enq = time.perf_counter() - t0          # host-side enqueue time (the GPU is still running)
        torch.cuda.synchronize()
        t_own = time.perf_counter() - t0
        if use_dist:
            dist.barrier()
        return time.perf_counter() - t0, t_own, t_enq, loss

    for _ in range(warmup):
        loss = step()
    comm_choice = None
    if use_dist and sync is not None and (world > 1 or os.environ.get("XG_FORCE_DIST") == "2"):
        # Which all-reduce schedule is faster HERE is decided by measurement, not by expectation: the overlapped buckets share the
        # CUs with the backward's chains and background products, and RCCL's kernels have never run beside them on more than one
        # rank on the build pool.  Three iterations each way (max over ranks), the faster one is timed; both are reported.
        tried = {}
        for name, ov in (("overlapped_buckets", True), ("one_collective_after_backward", False)):
            state["overlap_comm"] = ov
            step()
            t3, _, _, _ = timed(3)
            tt = torch.tensor([t3], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tried[name] = round(float(tt.item()) / 3 * 1e3, 3)
        state["overlap_comm"] = tried["overlapped_buckets"] <= tried["one_collective_after_backward"]
        comm_choice = {"ms_per_step_tried": tried, "timed": "overlapped_buckets" if state["overlap_comm"] else "one_collective_after_backward"}
    dt, dt_own, t_enq, loss = timed(steps)
    if use_dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    final_loss = float(loss.item())
    # ---- data-parallel diagnosis: the same iterations with the collective switched off (same streams, events and update
    # segments; only the RCCL kernels are missing) -> what the all-reduce costs each rank, i.e. its EXPOSED part
    comm = None
    if use_dist and comm_diag:
        try:
            tr._SKIP_COLLECTIVE = True
            try:
                for _ in range(2):
                    step()
                _, dt_nc, _, _ = timed(steps)
            finally:
                tr._SKIP_COLLECTIVE = False
            broadcast_parameters(model)           # (replicas diverged without the collective: back to rank 0's)
            # host side of THIS rank: the loop's wall time per iteration (enqueue + queue back-pressure) and the enqueue cost of one
            # iteration against an idle GPU -- with N ranks on one host these show at once whether a rank is host-bound
            enq_r = []
            for _ in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                step()
                enq_r.append(time.perf_counter() - t1)
            torch.cuda.synchronize()
            mine = {"rank": rank, "ms_per_step": round(dt_own / steps * 1e3, 3), "ms_per_step_no_collective": round(dt_nc / steps * 1e3, 3),
                    "exposed_comm_ms": round((dt_own - dt_nc) / steps * 1e3, 3),
                    "host_loop_ms_per_step": round(t_enq * 1e3 / steps, 3),
                    "host_enqueue_ms_per_step": round(sorted(enq_r)[1] * 1e3, 3),
                    "host_cores": core_ranges(ctx.get("cores"))}
            allr = [None] * world
            dist.all_gather_object(allr, mine)
            comm = {"per_rank": allr, "gradient_bytes": int(model.flat_grads().numel() * 4), "schedule": comm_choice,
                    "buckets": "logit | lstmcore+embed+img_embed | classifer | two_spatial_encoder (train.GradSync)"
                               if (sync is not None and state["overlap_comm"]) else "one",
                    "rccl_channels": rccl_channels(ctx.get("rccl_log")) if ctx.get("rccl_log") else None,
                    "rccl_channel_cap": ctx.get("rccl_cap"),
                    "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version())}
        except Exception as e:                    # never lose the scaling line over the diagnosis
            comm = {"error": repr(e)}
    # host-side enqueue cost of one iteration against an IDLE GPU (the loop above also contains queue back-pressure: the
    # host runs ahead of the GPU until the launch queue is full), median of 5
    enq = []
    for _ in range(5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        enq.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    host_enqueue_ms = sorted(enq)[len(enq) // 2] * 1e3
    host_enqueue_eager_ms = None
    if use_graph:                                # the same iteration as individual launches, for comparison
        enq = []
        for _ in range(5):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            eager_step()
            enq.append(time.perf_counter() - t1)
        torch.cuda.synchronize()
        host_enqueue_eager_ms = sorted(enq)[len(enq) // 2] * 1e3
    # in-situ duration of the T decoder steps inside the timed iteration (XgRun.prof_event0/1: recorded by the library on
    # the caller's stream around its time loop), median of 5 iterations
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record()
    model._prof_events = (e0, e1)
    insitu = []
    for _ in range(5):
        eager_step()                             # (events inside a replayed graph cannot be read: eager launches here)
        torch.cuda.synchronize()
        insitu.append(e0.elapsed_time(e1) * 1e3 / (T - 1 if workload == "scst" else T))   # (a rollout runs L core steps: the reference's L + 1st is dead)
    model._prof_events = None
    in_situ_us = sorted(insitu)[len(insitu) // 2]

    t_step = measure_step_group(model, x) if rank == 0 else None
    if rank != 0:
        return None, None
    step_rows = 2 * cfg["B"] if workload == "scst" else cfg["B"]      # (the SCST pair is ONE 2m-row batch)
    if workload == "scst":                       # the stand-alone group at the row count the rollout steps really have
        x2 = {k: torch.cat([v, v]) for k, v in x.items()}
        t_step = measure_step_group(model, x2)
    ms = dt / steps * 1e3
    value = world * cfg["B"] * (2 * (T - 1) if workload == "scst" else T) * steps / dt     # scst: L core steps per row actually run, two rollouts
    bf16 = precision == "bf16"
    bytes_step = step_bytes(step_rows, cfg["K"], cfg["R"], cfg["A"], cfg["E"], save=False, elem=2 if bf16 else 4)
    bytes_step_saved = step_bytes(step_rows, cfg["K"], cfg["R"], cfg["A"], cfg["E"], save=True, elem=2 if bf16 else 4)
    mfma_peak = 2500.0 if bf16 else 157.3
    achieved = bytes_step / t_step / 1e9
    traffic, traffic_src = (None, None)
    if cfg["B"] == 128 and workload == "xe" and precision == "fp32":
        if pmc and not args.no_pmc and world == 1:           # (the two rocprofv3 passes run on rank 0's GPU: single-GPU runs only)
            traffic, traffic_src = measure_traffic()
        elif world > 1:
            traffic_src = "multi-GPU run"
        if traffic is None:                              # fall back to the committed passes, and say so
            why = traffic_src
            traffic, traffic_src = load_traffic()
            if traffic_src:
                traffic_src += " (committed earlier: live PMC pass unavailable -- %s)" % why
    wl = {"xe": "configs[%s]: %dxMI355X batch %d per GPU teacher-forced XE train, 26 frames x (1536+1024), hidden 512, "
                "att 1536, emb 468, vocab 20000, seq_len 20 (T=21), fp32" % ("1" if world == 1 else "3", world, cfg["B"]),
          "scst": "configs[2]: %dxMI355X SCST iteration (sampled rollout + greedy baseline as one 2m-row batch + RL backward + "
                  "clip + Adam), batch %d per GPU, seq_len 30, 26 frames, hidden 512, vocab 20000, CIDEr reward stubbed" % (world, cfg["B"]),
          "xe5": "configs[4]: %dxMI355X batch %d per GPU teacher-forced XE train, 40 frames x (1536+1024), hidden 1024, att 1536, "
                 "vocab 20000, seq_len 20, %s" % (world, cfg["B"], precision)}[workload]
    flops = step_flops(step_rows, cfg["K"], cfg["R"], cfg["A"], cfg["E"])
    out = {
        "metric": "rollout timesteps/sec, SCST iteration (sample + greedy + RL backward) at batch 64, seq_len 30"
                  if workload == "scst" else "decoder timesteps/sec (train fwd+bwd) at batch 128, MSRVTT 26x1536+1024",
        "value": round(value, 1), "unit": "decoder timesteps/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": round(ms, 3), "host_enqueue_ms_per_step": round(host_enqueue_ms, 3),
        "host_loop_ms_per_step": round(t_enq * 1e3 / steps, 3),
        "host_enqueue_ms_per_step_eager": None if host_enqueue_eager_ms is None else round(host_enqueue_eager_ms, 3),
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": {"fp32": "f32", "bf16x3": "f32 via split-bf16 (3 planes, 6 MFMAs)", "bf16": "bf16"}[precision],
        "data": "synthetic",
        "config": {"workload": wl,
                   "global_batch": world * cfg["B"], "per_gpu_batch": cfg["B"], "parallelism": "dp%d" % world,
                   "path": args.path, "drop_prob_lm": drop, "gemm_precision": precision,
                   "timed_region": ("zero_grad + sampled + greedy rollouts (30 core steps, one 2m-row batch) + reward criterion + backward"
                                    if workload == "scst" else
                                    "zero_grad + encoder fwd + 21 decoder steps + heads/loss + full backward")
                                   + ((" + gloo grad all-reduce" if os.environ.get("XG_FORCE_DIST") == "3" else " + RCCL grad all-reduce") if world > 1 else "") + " + clip + Adam",
                   # (zero_grad is fused into the update: train.ClipAdam(fused_zero=True) leaves .grad at zero)
                   "zero_grad": "fused into the update",
                   "launch": "one HIP graph replay per iteration (train.GraphedXEStep)" if use_graph else "eager kernel launches",
                   "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")},
        "final_loss": round(final_loss, 5),
        **({"NOT_A_MEASUREMENT": "XG_FORCE_DIST=3: %d ranks time-slicing ONE GPU, all-reduce over gloo -- control-flow exercise only" % world}
           if os.environ.get("XG_FORCE_DIST") == "3" else {}),
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(achieved / 8000.0, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "decoder step launch group (xg_step_fwd: attention + POS gate + lstm_1 + lstm_2)",
                     "measured_form": "xg_step_fwd stand-alone, %d rows, inference form (tokens gathered inside the products, state "
                                      "updated in place, nothing saved), 200 back-to-back calls between two stream events" % step_rows,
                     "algorithmic_bytes_per_launch": bytes_step, "avg_launch_us": round(t_step * 1e6, 2),
                     # the same steps where the timed iteration runs them: %s loop of the forward pass (saves on, beside the
                     # side-stream products), events recorded by the library around its time loop
                     "in_situ_form": ("rollout loop (token choice + step + vocabulary product per step)" if workload == "scst"
                                      else "teacher-forced loop (token side hoisted, activations saved)") + ", inside the timed iteration",
                     "in_situ_us_per_step": round(in_situ_us, 2),
                     "in_situ_algorithmic_bytes": bytes_step_saved,
                     "in_situ_frac": None if workload == "scst" else round(bytes_step_saved / (in_situ_us * 1e-6) / 8e12, 4),
                     # the same launch group against the OTHER roof (exact-fp32 MFMA, 157.3 TF): at B = 128 the step's
                     # arithmetic intensity (33 FLOP/B) is above the ridge (20), i.e. the MFMA roof is the lower one
                     "mfma_tflops": round(flops / t_step / 1e12, 2),
                     "mfma_peak_tflops": mfma_peak,
                     "mfma_frac": round(flops / t_step / (mfma_peak * 1e12), 4)},
    }
    if workload != "scst":
        # the WHOLE iteration against the matrix roof: chains, products and the update share the chip for dt / steps seconds
        it_flop = iteration_flops(cfg)
        out["roofline"]["iteration"] = {"bound": "mfma", "flop": it_flop, "achieved": round(it_flop / (dt / steps) / 1e12, 1),
                                        "peak": mfma_peak, "unit": "TFLOP/s", "frac": round(it_flop / (dt / steps) / (mfma_peak * 1e12), 4),
                                        "what": "2 x multiply-adds of one training iteration (forward + twice that for the backward) over "
                                                "ms_per_step, against the dense MFMA peak of the products' arithmetic"}
    if comm is not None:
        out["comm"] = comm
    parity_fail = None
    if world == 1 and cpu_leg and not args.no_cpu_baseline:
        tol = 1e-2 if precision == "bf16" else 1e-4        # north_star: 1e-4 on the training loss in fp32, 1e-2 for the bf16 config
        if workload == "scst":
            par = scst_parity(model, cfg, x, reward_b)
            what = "RewardCriterion loss of the HIP paired rollout vs the CPU oracle replaying the tokens it drew (same weights, same batch)"
        else:
            cb = cpu_baseline(cfg, args.cpu_budget if (workload == "xe" and drop == args.drop) else 0.0, model, x, p=drop, seed=20240605)
            par = cb.pop("parity", None)
            out["cpu_baseline"] = cb
            what = "XE loss of this run's HIP model vs the CPU oracle, same procedural weights, same batch" + \
                   (" and the same dropout masks (p = %g, shared integer-hash seed)" % drop if drop > 0.0 else "")
        if par is not None:
            out["parity_loss_delta"] = round(par["delta"], 7)
            out["parity"] = {"hip_loss": round(par["hip_loss"], 6), "oracle_loss": round(par["oracle_loss"], 6), "tol": tol, "what": what}
            if not par["delta"] < tol:
                parity_fail = "parity (%s %s): |hip - oracle| = %.3g >= %g" % (workload, precision, par["delta"], tol)
    return out, parity_fail


def step_group_by_arithmetic(args, ctx):
    """The stand-alone decoder-step launch group (xg_step_fwd, 128 rows, hidden 512 -- the roofline kernel of the headline line)
    in each arithmetic mode of the per-step products: exact fp32 MFMA, split-bf16 (three planes, six bf16 MFMAs per 16-deep
    block: fp32-class results) and plain bf16.  Same weights, same inputs; microseconds per launch group."""
    from controllable_xgating_amd import SAModel, make_opt
    cfg = dict(B=128, K=26, R=512, A=1536, E=468, V=20000, C=14, L=20, F1=1536, F2=1024)
    x = synth_inputs(cfg["B"], cfg["K"], cfg["L"], cfg["V"], cfg["R"], cfg["F1"], cfg["F2"], cfg["C"], 0, ctx["dev"])
    out = {}
    for precision in ("fp32", "bf16x3", "bf16"):
        torch.manual_seed(0)
        opt = make_opt(None, vocab_size=cfg["V"], seq_length=cfg["L"], precision=precision, rnn_size=cfg["R"], att_size=cfg["A"],
                       input_encoding_size=cfg["E"], feat_size=cfg["F1"], feat_size2=cfg["F2"])
        model = SAModel(opt).to(ctx["dev"])
        model.eval()
        out[precision] = round(measure_step_group(model, x) * 1e6, 2)
        del model
    out["what"] = ("xg_step_fwd stand-alone, 128 rows, hidden 512, 200 back-to-back calls; fp32 = v_mfma_f32_32x32x2_f32 on packed fp32 "
                   "tiles; bf16x3 = the same weights as three pre-split bf16 planes in the packed tiles (packed_dtype 2), activations split "
                   "while staged, 6 x v_mfma_f32_32x32x16_bf16 per 16-deep block (0.375 of the fp32 matrix time); bf16 = packed bf16 "
                   "tiles, 1 MFMA per block")
    return out


def main():
    # stdout carries exactly ONE line, the JSON: everything else that writes to file descriptor 1 (RCCL prints a version banner
    # through C stdio when a communicator comes up, rocprofv3 children, ...) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--path", choices=["fused", "surface"], default="fused",
                    help="fused: model.xe_loss (no (m,T,V) gradient tensor); surface: model() + criterion, reference call sequence")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--drop", type=float, default=0.0)
    ap.add_argument("--precision", choices=["fp32", "bf16x3", "bf16"], default="fp32",
                    help="arithmetic of the large GEMMs (XgRun.gemm_mode); the headline number is fp32")
    ap.add_argument("--workload", choices=["xe", "scst", "xe5"], default="xe",
                    help="xe: BASELINE configs[1] (the metric); scst: configs[2] (sample + greedy rollouts + RL backward, B=64, L=30); "
                         "xe5: configs[4] shape (hidden 1024, 40 frames; pair with --precision bf16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc passes that measure roofline.traffic")
    ap.add_argument("--graph", action="store_true", help="time the HIP-graph replay of the iteration (single-stream capture) instead of eager launches")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the configs[2] (SCST) and configs[4] (hidden-1024 bf16) lines that the default single-GPU run appends")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU, RCCL over xGMI) -- never
        # silently run one GPU and call it N
        ndev = torch.cuda.device_count()
        if ndev < args.gpus and os.environ.get("XG_FORCE_DIST") != "3":
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" % (args.gpus, ndev))
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.dup2(json_fd, 1)                      # (the launched ranks do their own redirection)
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # XG_FORCE_DIST=3: the N-rank control flow of this file on ONE GPU -- every rank on device 0, gloo instead of RCCL (which refuses
    # two ranks on one device).  Exercises the sharding, the barriers, the max over ranks, the per-rank report and the collective's
    # call sites where no multi-GPU box exists; its numbers mean nothing (time-sliced GPU, host-staged all-reduce) and say so.
    one_gpu = os.environ.get("XG_FORCE_DIST") == "3"
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("XG_FORCE_DIST") in ("1", "2")    # XG_FORCE_DIST=1|2: exercise the RCCL path on one GPU
    ctx = dict(world=world, rank=rank, dev=dev, use_dist=use_dist, rccl_log=None, cores=None, rccl_cap=None)
    if use_dist:
        ctx["cores"] = pin_host_thread(local, world)
        ctx["rccl_cap"] = cap_rccl_channels()
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        ctx["rccl_log"] = rccl_debug_setup(rank)
        from controllable_xgating_amd.train import dist_diagnosis
        try:
            if one_gpu:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)                  # the first collective brings the communicator up: fail HERE, with the one line that explains it
            torch.cuda.synchronize()
            assert int(probe.item()) == world, probe
        except Exception:
            print(dist_diagnosis(), file=sys.stderr, flush=True)
            raise
        if rank == 0:
            print(dist_diagnosis(), file=sys.stderr, flush=True)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if use_dist:
        dist.barrier()

    out, fail = run_workload(args, args.workload, args.precision, args.steps, args.warmup, ctx)
    fails = [fail] if fail else []
    # the default single-GPU run also carries BASELINE.json's other single-GPU configurations (5 warm-up + 10 timed
    # iterations each, own roofline and parity self-check): configs[2] (SCST) and configs[4] (hidden 1024 / 40 frames, bf16)
    if (not args.no_secondary and world == 1 and not use_dist and args.workload == "xe" and args.precision == "fp32"
            and args.batch == 128 and args.path == "fused"):
        sec = {}
        # ... and the headline configuration at the reference's default drop_prob_lm = 0.5 (myopts.py:37; SURVEY.md 8d: "throughput
        # reported at 0.0 and 0.5"): secondary.xe_drop05
        for key, (wl_, prec_, drop_) in {"scst": ("scst", "fp32", None), "xe5_bf16": ("xe5", "bf16", None), "xe_bf16x3": ("xe", "bf16x3", None),
                                         "xe_drop05": ("xe", "fp32", 0.5)}.items():
            if drop_ is not None and drop_ == args.drop:
                continue
            try:
                o, f = run_workload(args, wl_, prec_, 10, 5, ctx, pmc=False, comm_diag=False, drop=drop_)
                sec[key] = {k: o[k] for k in ("metric", "value", "unit", "ms_per_step", "host_enqueue_ms_per_step", "dtype", "config",
                                              "final_loss", "roofline", "parity_loss_delta", "parity", "steps", "warmup") if k in o}
                if "cpu_baseline" in o:
                    sec[key]["cpu_baseline"] = o["cpu_baseline"]
                if f:
                    fails.append(f)
            except Exception as e:                       # never lose the headline over a secondary line
                sec[key] = {"error": repr(e)}
                fails.append("secondary %s failed: %r" % (key, e))
        out["secondary"] = sec
        # one shape, three arithmetic modes: the decoder-step launch group of the headline config (128 rows, hidden 512)
        try:
            out["roofline"]["step_us_by_arithmetic"] = step_group_by_arithmetic(args, ctx)
        except Exception as e:
            out["roofline"]["step_us_by_arithmetic"] = {"error": repr(e)}
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
        if fails:
            raise SystemExit("; ".join(fails))


if __name__ == "__main__":
    main()
