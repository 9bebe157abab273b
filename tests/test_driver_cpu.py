"""Schedules of the train-driver counterpart against values worked out from
reference caption_src/starttrain.py:84-107 and caption_src/myopts.py defaults.  CPU only."""
import argparse

import json
import os

import numpy as np
import torch

from controllable_xgating_amd.driver import decode_sequence, lr_for_epoch, sc_flag_for_epoch, ss_prob_for_epoch


def _opt(**kw):
    base = dict(learning_rate=4e-4, learning_rate_decay_start=0, learning_rate_decay_every=3, learning_rate_decay_rate=0.5,
                scheduled_sampling_start=3, scheduled_sampling_increase_every=5, scheduled_sampling_increase_prob=0.05,
                scheduled_sampling_max_prob=0.25, self_critical_after=30)
    base.update(kw)
    return argparse.Namespace(**base)


def test_lr_step_decay():
    o = _opt()
    assert lr_for_epoch(o, 0) == 4e-4                       # epoch > start is strict (starttrain.py:88)
    assert lr_for_epoch(o, 2) == 4e-4                       # int(2/3) = 0
    assert abs(lr_for_epoch(o, 3) - 2e-4) < 1e-12
    assert abs(lr_for_epoch(o, 7) - 1e-4) < 1e-12           # int(7/3) = 2
    assert lr_for_epoch(_opt(learning_rate_decay_start=-1), 50) == 4e-4


def test_scheduled_sampling_ramp():
    o = _opt()
    assert ss_prob_for_epoch(o, 3) == 0.0                   # strict >
    assert ss_prob_for_epoch(o, 7) == 0.0                   # int(4/5) = 0
    assert abs(ss_prob_for_epoch(o, 8) - 0.05) < 1e-12
    assert abs(ss_prob_for_epoch(o, 100) - 0.25) < 1e-12    # capped at max_prob
    assert ss_prob_for_epoch(_opt(scheduled_sampling_start=-1), 9, current=0.1) == 0.1


def test_self_critical_switch():
    o = _opt()
    assert not sc_flag_for_epoch(o, 29) and sc_flag_for_epoch(o, 30)
    assert not sc_flag_for_epoch(_opt(self_critical_after=-1), 1000)


def test_decode_sequence_vs_reference_fixture():
    """myutils.decode_sequence (myutils.py:88-102): fixture recorded from the reference's own function
    (tools/gen_golden.py:gen_decode): rows cut at the first 0, an empty row, a row without 0, tokens after a 0 ignored."""
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "decode_seq.json")))
    vocab_int = {int(k): v for k, v in g["vocab"].items()}
    seq = np.asarray(g["seq"], dtype=np.int64)
    assert decode_sequence(vocab_int, torch.from_numpy(seq)) == g["sents"]
    assert decode_sequence(g["vocab"], seq) == g["sents"]            # json vocabularies have str keys
