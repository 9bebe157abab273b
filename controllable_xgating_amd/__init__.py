"""controllable_xgating_amd -- MI355X-native (gfx950) hot path of the gated-fusion caption decoder
(reference: vsislab/Controllable_XGating, caption_src/SAModel.py).  See DESIGN.md."""
import os as _os

# One training process uses up to seven HIP streams (the caller's, the library's two side streams, the optimizer's and the
# gradient all-reduce's side streams, RCCL's own).  The ROCm runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware
# queues (default 4): beyond that, independent streams share a queue and serialise -- measured with the data-parallel path
# forced on one GPU (XG_FORCE_DIST=1): 7.44 ms per iteration with 4 queues, 6.25 with 6, 6.18 with 8 (6.10 without the
# collective's streams).  Must be in the environment before the HIP runtime starts; a value the user set wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from ._native import XgError, lib, LIB_PATH  # noqa: F401,E402
from .model import (SAModel, LanguageModelCriterion, ClassiferCriterion, RewardCriterion, make_opt)  # noqa: F401,E402

__all__ = ["SAModel", "LanguageModelCriterion", "ClassiferCriterion", "RewardCriterion", "make_opt", "XgError", "lib"]
