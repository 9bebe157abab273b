#!/usr/bin/env python3
"""Runs greedy + sampled rollouts at config 3 shape (B=64, L=30) for profiling."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from controllable_xgating_amd import SAModel, make_opt
B, L = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 30
model = SAModel(make_opt(None, vocab_size=20000, seq_length=L)).cuda(); model.train()
x = bench.synth_inputs(B, 26, L, 20000, 512, 1536, 1024, 14, 0, "cuda")
for _ in range(6):
    with torch.no_grad():
        model.sample(x["feats_rgb"], x["feats_opfl"], x["feat_mask"], x["pos_feats"], {"sample_max": 1})
torch.cuda.synchronize()
