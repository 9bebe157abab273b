"""Schedules of the train-driver counterpart against values worked out from
reference caption_src/starttrain.py:84-107 and caption_src/myopts.py defaults.  CPU only."""
import argparse

from controllable_xgating_amd.driver import lr_for_epoch, sc_flag_for_epoch, ss_prob_for_epoch


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
