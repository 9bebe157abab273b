"""controllable_xgating_amd -- MI355X-native (gfx950) hot path of the gated-fusion caption decoder
(reference: vsislab/Controllable_XGating, caption_src/SAModel.py).  See DESIGN.md."""
from ._native import XgError, lib, LIB_PATH  # noqa: F401
from .model import (SAModel, LanguageModelCriterion, ClassiferCriterion, RewardCriterion, make_opt)  # noqa: F401

__all__ = ["SAModel", "LanguageModelCriterion", "ClassiferCriterion", "RewardCriterion", "make_opt", "XgError", "lib"]
