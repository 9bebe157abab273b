#!/usr/bin/env python3
"""Runs ONLY the decoder-step launch group (xg_step_fwd) of bench.py's config, for rocprofv3 PMC passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from controllable_xgating_amd import SAModel, make_opt

cfg = dict(B=128, K=26, R=512, A=1536, E=468, V=20000, C=14, L=20, F1=1536, F2=1024)
model = SAModel(make_opt(None, vocab_size=cfg["V"], seq_length=cfg["L"])).cuda()
model.train()
x = bench.synth_inputs(cfg["B"], cfg["K"], cfg["L"], cfg["V"], cfg["R"], cfg["F1"], cfg["F2"], cfg["C"], 0, "cuda")
t = bench.measure_step_group(model, x, reps=int(sys.argv[1]) if len(sys.argv) > 1 else 40)
print("step group %.2f us" % (t * 1e6))
