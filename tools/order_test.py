import sys, os, gc, argparse, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, bench
torch.cuda.set_device(0)
import __graft_entry__ as ge; ge.build()
args = argparse.Namespace(batch=128, drop=0.0, path="fused", no_pmc=True, no_cpu_baseline=True, cpu_budget=1.0, graph=bool(os.environ.get("GRAPH")))
ctx = dict(world=1, rank=0, dev=torch.device("cuda", 0), use_dist=False, rccl_log=None)
def run(wl, prec):
    o, _ = bench.run_workload(args, wl, prec, 10, 5, ctx, cpu_leg=False, pmc=False, comm_diag=False)
    gc.collect()
    return o["ms_per_step"], o["roofline"]["in_situ_us_per_step"]
seq = sys.argv[1].split(",")
for s in seq:
    wl, prec = s.split(":")
    print(s, run(wl, prec), flush=True)
