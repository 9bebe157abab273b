#!/usr/bin/env python3
"""Eight launcher processes on ONE host (SURVEY.md 8e readiness that needs no 8-GPU node): every process pins its host thread to
its own slice of the allowed cores (as bench.py --gpus 8 does per LOCAL_RANK), builds its own model and runs the eager XE
iteration loop -- all of them against the one visible GPU, i.e. their kernels time-slice it, which is fine: the question is the
HOST side.  Reported per process: the wall time its Python / HIP-runtime launch loop needs to ENQUEUE one iteration (~300
launches on four streams) while seven other launchers do the same on the same host.  If those numbers stay near the single-process
figure (bench.py: host_enqueue_ms_per_step, ~2 ms) the hosts's launch paths do not collide; the GPU-side times printed next to
them are NOT scaling figures (one GPU is shared by all eight).

Two passes: with the product library (the enqueue time then contains whatever the HIP runtime makes a launcher WAIT for when eight
processes time-slice one GPU: the 43-76 ms maxima of round 5) and with the -DXG_NULL_LAUNCH build (every launch an empty kernel:
the same call sequence with no GPU time behind it = the host's own cost; build it first:
python -c "import __graft_entry__ as g; g.build_variant('null', ['-DXG_NULL_LAUNCH'])").

usage: host8_enqueue.py [nproc (8)] [iterations (6)]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, iters, q):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import bench
    cores = bench.pin_host_thread(rank, world)
    from controllable_xgating_amd import SAModel, make_opt
    from controllable_xgating_amd.train import ClipAdam
    cfg = dict(B=128, K=26, R=512, A=1536, E=468, V=20000, C=14, L=20, F1=1536, F2=1024)
    dev = torch.device("cuda", 0)
    model = SAModel(make_opt(None, vocab_size=cfg["V"], seq_length=cfg["L"])).to(dev)
    model.train()
    x = bench.synth_inputs(cfg["B"], cfg["K"], cfg["L"], cfg["V"], cfg["R"], cfg["F1"], cfg["F2"], cfg["C"], seed=rank, device=dev)
    optim = ClipAdam(model, lr=4e-4, grad_clip=0.1, overlap=True, fused_zero=True)

    def step():
        optim.zero_grad()
        loss = model.xe_loss(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], x["seq"], x["seq_mask"])
        optim.arm()
        loss.backward()
        optim.step()
        return loss
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    q.put(("ready", rank))
    while not os.path.exists(os.path.join(os.environ["H8_DIR"], "go")):
        time.sleep(0.001)
    enq, tot = [], []
    for _ in range(iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
    q.put(("done", rank, sorted(enq)[len(enq) // 2], min(enq), max(enq), sorted(tot)[len(tot) // 2], bench.core_ranges(cores) if cores else None))


def main():
    import multiprocessing as mp
    import tempfile
    nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    ctx = mp.get_context("spawn")
    out = {}
    null_lib = os.path.join(ROOT, "controllable_xgating_amd", "lib", "libxgate_hip_null.so")
    passes = [("product_library", None)] + ([("null_launch_library", null_lib)] if os.path.exists(null_lib) else [])
    for label, lib in passes:
      if lib:
          os.environ["XG_LIBRARY"] = lib
      for world in (1, nproc):
          d = tempfile.mkdtemp(prefix="h8_")
          os.environ["H8_DIR"] = d
          q = ctx.Queue()
          ps = [ctx.Process(target=worker, args=(r, world, iters, q)) for r in range(world)]
          for p in ps:
              p.start()
          ready = 0
          while ready < world:
              m = q.get(timeout=600)
              ready += m[0] == "ready"
          open(os.path.join(d, "go"), "w").close()
          res = []
          while len(res) < world:
              m = q.get(timeout=600)
              if m[0] == "done":
                  res.append(m[1:])
          for p in ps:
              p.join()
          res.sort()
          out.setdefault(label, {})["%d_process%s" % (world, "es" if world > 1 else "")] = [
              {"rank": r[0], "host_enqueue_ms_per_step_median": round(r[1], 3), "min": round(r[2], 3), "max": round(r[3], 3),
               "iteration_ms_shared_gpu": round(r[4], 3), "host_cores": r[5]} for r in res]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
