"""Data-parallel HIP path on TWO MI355X (skipped when the box has one GPU): one process per GPU, RCCL.
Parity is defined per shard: each rank's HIP gradients equal the oracle's on that rank's videos (r::world), and after the
one gradient all-reduce (plain, and the overlapped two-bucket GradSync form) every rank holds the mean of the per-shard
oracle gradients.  starttrain.py:125-137 is single-GPU; this is the `north_star`'s shard-by-video extension."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    from controllable_xgating_amd import train as tr
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
        probe = torch.ones(1, device="cuda:%d" % rank)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        assert int(probe.item()) == world
    except Exception:
        print(tr.dist_diagnosis(), flush=True)          # the one line that explains a first two-GPU failure
        raise
    from oracle import paramgen as pg
    from oracle import xgate_oracle as xo
    from tests.util import CFG, ZERO_GRAD_PARAMS, make_model
    torch.set_num_threads(4)
    d = pg.make_dims(**CFG["mid"])
    Pn = pg.make_params(d)
    x_all = xo.to_torch_inputs(pg.make_inputs(d, seed=0, ragged=True))
    xs = tr.shard_batch(x_all, rank, world)
    # oracle on this rank's shard
    P = xo.to_torch_params(Pn, requires_grad=True)
    logp, _, _ = xo.forward_xe(P, xs["feats_rgb"], xs["feats_opfl"], xs["feat_mask"], xs["pos_feats"], xs["seq"], xs["seq_mask"],
                               train=True, running=xo.new_running(d))
    loss_o = xo.lm_criterion(logp, xs["seq"], xs["seq_mask"])
    loss_o.backward()
    names = list(P)
    g_o = {n: (P[n].grad if P[n].grad is not None else torch.zeros_like(P[n])).numpy() for n in names}
    # every rank needs every shard's oracle gradient for the mean: exchange them through the collective under test's peer, gloo-free
    flat_o = torch.cat([torch.from_numpy(g_o[n]).reshape(-1) for n in names]).cuda()
    mean_o = flat_o.clone()
    dist.all_reduce(mean_o, op=dist.ReduceOp.SUM)
    mean_o = (mean_o / world).cpu().numpy()
    xd = {k: v.cuda() for k, v in xs.items()}
    res = {}
    for mode in ("plain", "gradsync"):
        model = make_model(d, P=Pn, device="cuda:%d" % rank)
        if rank != 0:                                  # replicas start from rank 0's weights (broadcast_parameters)
            with torch.no_grad():
                model.flat_parameters().zero_()
        tr.broadcast_parameters(model)
        sync = tr.GradSync(model) if mode == "gradsync" else None
        model.flat_grads().zero_()
        loss = model.xe_loss(xd["feats_rgb"], xd["feats_opfl"], xd["feat_mask"], xd["pos_feats"], xd["seq"], xd["seq_mask"])
        if sync is not None:
            sync.arm()
        loss.backward()
        torch.cuda.synchronize()
        assert abs(loss.item() - loss_o.item()) < 1e-4, (rank, loss.item(), loss_o.item())
        if sync is None:                               # local gradients == oracle on the shard (before any collective)
            for n, prm in model.named_parameters():
                if n in ZERO_GRAD_PARAMS:
                    continue
                err = np.abs(prm.grad.cpu().numpy() - g_o[n]).max()
                assert err <= 2e-6 + 2e-3 * np.abs(g_o[n]).max(), (rank, n, float(err))
        tr.allreduce_gradients(model)
        torch.cuda.synchronize()
        got = torch.cat([dict(model.named_parameters())[n].grad.reshape(-1) for n in names]).cpu().numpy()
        res[mode] = got
        off = 0
        for n in names:
            k = g_o[n].size
            if n not in ZERO_GRAD_PARAMS:
                ref = mean_o[off:off + k]
                err = np.abs(got[off:off + k] - ref).max()
                assert err <= 2e-6 + 2e-3 * np.abs(ref).max(), (mode, rank, n, float(err))
            off += k
    np.save(os.path.join(out_dir, "g%d.npy" % rank), res["plain"])
    np.save(os.path.join(out_dir, "s%d.npy" % rank), res["gradsync"])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_hip_data_parallel_parity(tmp_path):
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (the driver's 1-GPU box skips this; covered on CPU by tests/test_dp_gloo.py)")
    import __graft_entry__ as ge
    ge.build()
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy")
    assert np.array_equal(g0, g1)                      # every replica holds the same averaged gradient
    s0, s1 = np.load(tmp_path / "s0.npy"), np.load(tmp_path / "s1.npy")
    assert np.array_equal(s0, s1)
    np.testing.assert_allclose(s0, g0, rtol=2e-3, atol=2e-6)


def _worker_update(rank, world, port, out_dir):
    """Three data-parallel training iterations per update path; every path must leave the same parameters / moments on
    every rank: (a) plain all-reduce + stream-ordered ClipAdam, (b) plain all-reduce + ARMED overlapping ClipAdam(fused_zero)
    (allreduce_gradients disarms it), (c) GradSync.finish + armed overlapping ClipAdam (segment updates behind each bucket)."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from controllable_xgating_amd import train as tr
    from oracle import paramgen as pg
    from oracle import xgate_oracle as xo
    from tests.util import CFG, make_model
    d = pg.make_dims(**CFG["mid"])
    Pn = pg.make_params(d)
    xs = tr.shard_batch(xo.to_torch_inputs(pg.make_inputs(d, seed=0, ragged=True)), rank, world)
    xd = {k: v.cuda() for k, v in xs.items()}
    for mode in ("plain", "plain_overlap", "gradsync_overlap"):
        model = make_model(d, P=Pn, device="cuda:%d" % rank)
        tr.broadcast_parameters(model)
        over = mode != "plain"
        opt = tr.ClipAdam(model, lr=4e-4, grad_clip=0.1, overlap=over, fused_zero=over)
        sync = tr.GradSync(model) if mode == "gradsync_overlap" else None
        for _ in range(3):
            opt.zero_grad()
            loss = model.xe_loss(xd["feats_rgb"], xd["feats_opfl"], xd["feat_mask"], xd["pos_feats"], xd["seq"], xd["seq_mask"])
            if sync is not None:
                sync.arm()
            opt.arm()
            loss.backward()
            tr.allreduce_gradients(model)
            opt.step()
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, "%s_p%d.npy" % (mode, rank)), model.flat_parameters().detach().cpu().numpy())
        np.save(os.path.join(out_dir, "%s_m%d.npy" % (mode, rank)), opt.exp_avg.cpu().numpy())
        np.save(os.path.join(out_dir, "%s_v%d.npy" % (mode, rank)), opt.exp_avg_sq.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_update_paths_agree(tmp_path):
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (single-rank mechanics: tests/test_gpu_parity.py::test_data_parallel_update_paths_agree_single_rank)")
    import __graft_entry__ as ge
    ge.build()
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker_update, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ref = {k: np.load(tmp_path / ("plain_%s0.npy" % k)) for k in "pmv"}
    for mode in ("plain", "plain_overlap", "gradsync_overlap"):
        for k in "pmv":
            a, b = np.load(tmp_path / ("%s_%s0.npy" % (mode, k))), np.load(tmp_path / ("%s_%s1.npy" % (mode, k)))
            assert np.array_equal(a, b), (mode, k)                 # replicas stay bit-identical
        np.testing.assert_allclose(np.load(tmp_path / ("%s_m0.npy" % mode)), ref["m"], atol=1e-6 + 2e-3 * np.abs(ref["m"]).max(), err_msg=mode)
        np.testing.assert_allclose(np.load(tmp_path / ("%s_v0.npy" % mode)), ref["v"], atol=1e-9 + 2e-3 * np.abs(ref["v"]).max(), err_msg=mode)
        disp = np.abs(np.load(tmp_path / ("%s_p0.npy" % mode)) - ref["p"])
        assert disp.max() <= 3.1 * 4e-4 and (disp > 4e-5).mean() <= 0.02, (mode, float(disp.max()))


def test_bench_n_rank_control_flow_on_one_gpu_over_gloo(tmp_path):
    """`bench.py --gpus 2` end to end where only one GPU exists (XG_FORCE_DIST=3: both ranks on device 0, gloo instead of RCCL): the
    self-launch through torch.distributed.run, the shard per rank, barriers, max over ranks, schedule choice, per-rank report and
    the ONE JSON line from rank 0.  The numbers are not measurements (the line says so); what is asserted is the contract's shape."""
    import json, subprocess, sys
    from tests.util import ROOT
    env = dict(os.environ, XG_FORCE_DIST="3", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-pmc"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # rank 0 alone prints, once
    o = json.loads(lines[0])
    assert o["n_gpus"] == 2 and o["steps"] == 2 and o["scaling"] == "weak" and "NOT_A_MEASUREMENT" in o
    assert o["config"]["global_batch"] == 256 and o["config"]["parallelism"] == "dp2"
    assert abs(o["value"] - 2 * 128 * 21 / (o["ms_per_step"] * 1e-3)) <= 1e-3 * o["value"]      # whole-job aggregate over both ranks
    ranks = o["comm"]["per_rank"]
    assert [p["rank"] for p in ranks] == [0, 1] and all(p["ms_per_step"] > 0 and p["ms_per_step_no_collective"] > 0 for p in ranks)
    assert o["comm"]["gradient_bytes"] == 144489216 and o["comm"]["schedule"]["timed"] in ("overlapped_buckets", "one_collective_after_backward")
