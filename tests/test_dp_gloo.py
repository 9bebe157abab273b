"""Data-parallel path on CPU with gloo, world_size 2: rank r takes videos r::world, ONE
all-reduce(sum)/world of the flat gradient before the clamp, nothing else.  The per-rank gradients come from the
oracle on that rank's shard; the all-reduced result must equal the mean of the per-shard oracle gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import paramgen as pg
from oracle import xgate_oracle as xo
from tests.util import CFG


class _FlatHolder:
    """Stands in for SAModel's flat-buffer interface (flat_grads / flat_parameters / buffers)."""

    def __init__(self, g, p):
        self._g, self._p = g, p

    def flat_grads(self):
        return self._g

    def flat_parameters(self):
        return self._p

    def buffers(self):
        return []


def _shard_grads(d, rank, world):
    from controllable_xgating_amd.train import shard_batch
    P = xo.to_torch_params(pg.make_params(d), requires_grad=True)
    x = shard_batch(xo.to_torch_inputs(pg.make_inputs(d, seed=0)), rank, world)   # full-width captions on every shard
    logp, cat, _ = xo.forward_xe(P, x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], x["seq"], x["seq_mask"],
                                 train=True, running=xo.new_running(d))
    xo.lm_criterion(logp, x["seq"], x["seq_mask"]).backward()
    return torch.cat([(P[n].grad if P[n].grad is not None else torch.zeros_like(P[n])).reshape(-1) for n in P])


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from controllable_xgating_amd.train import allreduce_gradients, broadcast_parameters
    torch.set_num_threads(2)
    d = pg.make_dims(**CFG["tiny"])
    g = _shard_grads(d, rank, world)
    p = torch.full((8,), float(rank + 1))
    holder = _FlatHolder(g.clone(), p)
    allreduce_gradients(holder)
    broadcast_parameters(holder, src=0)
    np.save(os.path.join(out_dir, f"g{rank}.npy"), holder.flat_grads().numpy())
    np.save(os.path.join(out_dir, f"raw{rank}.npy"), g.numpy())
    np.save(os.path.join(out_dir, f"p{rank}.npy"), p.numpy())
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def test_allreduce_equals_mean_of_shard_gradients(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g0, g1 = np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy")
    raw = [np.load(tmp_path / f"raw{r}.npy") for r in range(world)]
    assert np.array_equal(g0, g1)                                  # every replica holds the same averaged gradient
    np.testing.assert_allclose(g0, (raw[0] + raw[1]) / 2, rtol=0, atol=1e-7)
    assert not np.allclose(raw[0], raw[1])                         # the shards really differed
    assert np.array_equal(np.load(tmp_path / "p1.npy"), np.load(tmp_path / "p0.npy"))   # broadcast from rank 0


def test_shard_batch_is_a_partition():
    from controllable_xgating_amd.train import shard_batch
    d = pg.make_dims(**CFG["tiny"])
    x = xo.to_torch_inputs(pg.make_inputs(d, seed=0))
    parts = [shard_batch(x, r, 2) for r in range(2)]
    assert parts[0]["seq"].shape[0] + parts[1]["seq"].shape[0] == d.B
    merged = torch.empty_like(x["feats_rgb"])
    merged[0::2], merged[1::2] = parts[0]["feats_rgb"], parts[1]["feats_rgb"]
    assert torch.equal(merged, x["feats_rgb"])


def test_single_process_is_a_noop():
    from controllable_xgating_amd.train import allreduce_gradients
    g = torch.arange(5.0)
    allreduce_gradients(_FlatHolder(g, g))                         # no process group: must not touch anything
    assert torch.equal(g, torch.arange(5.0))


# ---------------------------------------------------------------- world size 8 (BASELINE.json configs[3]: one node, 8 ranks)
def _worker8(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from controllable_xgating_amd.train import _reduce, allreduce_gradients, bucket_plan, shard_batch
    torch.set_num_threads(1)
    d = pg.make_dims(**dict(CFG["tiny"], B=19))                 # 19 videos over 8 ranks: shards of 3, 3, 3, 2, 2, 2, 2, 2
    x = xo.to_torch_inputs(pg.make_inputs(d, seed=0))
    mine = shard_batch(x, rank, world)
    np.save(os.path.join(out_dir, f"rows{rank}.npy"), mine["feats_rgb"].numpy())
    # a flat "gradient" that depends on the rank, reduced (a) as ONE collective and (b) as GradSync's buckets in plan order
    numel, split, head = 5000, 1300, (3100, 4420)
    base = torch.from_numpy(pg.uniform("dp8.g", (numel,), 3, -1.0, 1.0))
    g_one = base * (rank + 1)
    g_bkt = g_one.clone()
    allreduce_gradients(_FlatHolder(g_one, g_one))
    plan = bucket_plan(split, head, numel)
    for lo, hi, _ in plan:
        _reduce(g_bkt[lo:hi], world, None)
    np.save(os.path.join(out_dir, f"one{rank}.npy"), g_one.numpy())
    np.save(os.path.join(out_dir, f"bkt{rank}.npy"), g_bkt.numpy())
    with open(os.path.join(out_dir, f"plan{rank}.txt"), "w") as f:
        f.write(repr(plan))
    dist.destroy_process_group()


def test_world8_shards_partition_the_batch_and_buckets_are_deterministic(tmp_path):
    """Eight ranks (gloo): (1) shard_batch r::8 of a batch that does not divide by 8 is a partition, in rank-strided order;
    (2) every rank derives the SAME bucket plan (bounds and order), the buckets tile the flat buffer exactly once, and reducing
    bucket by bucket in plan order gives bit-identical results on all ranks, equal to the single-collective path."""
    from controllable_xgating_amd.train import bucket_plan
    world = 8
    mp.spawn(_worker8, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    d = pg.make_dims(**dict(CFG["tiny"], B=19))
    x = pg.make_inputs(d, seed=0)
    rows = [np.load(tmp_path / f"rows{r}.npy") for r in range(world)]
    assert [len(r) for r in rows] == [3, 3, 3, 2, 2, 2, 2, 2]
    merged = np.empty_like(x["feats_rgb"])
    for r in range(world):
        merged[r::world] = rows[r]
    assert np.array_equal(merged, x["feats_rgb"])
    plans = [open(tmp_path / f"plan{r}.txt").read() for r in range(world)]
    assert len(set(plans)) == 1
    plan = bucket_plan(1300, (3100, 4420), 5000)
    assert repr(plan) == plans[0] and [e for _, _, e in plan] == ["head", "rest", "rest", "end"]
    cover = np.zeros(5000, np.int32)
    for lo, hi, _ in plan:
        assert 0 <= lo < hi <= 5000
        cover[lo:hi] += 1
    assert (cover == 1).all()
    one = [np.load(tmp_path / f"one{r}.npy") for r in range(world)]
    bkt = [np.load(tmp_path / f"bkt{r}.npy") for r in range(world)]
    base = pg.uniform("dp8.g", (5000,), 3, -1.0, 1.0)
    for r in range(world):
        assert np.array_equal(one[r], one[0]) and np.array_equal(bkt[r], bkt[0])
    # (gloo's ring sums in a rank-dependent association per chunk: the two paths agree to round-off, every rank bit-identically)
    np.testing.assert_allclose(bkt[0], one[0], rtol=0, atol=2e-6)
    np.testing.assert_allclose(one[0], base * 4.5, rtol=0, atol=2e-6)        # mean of (r + 1) over 8 ranks
