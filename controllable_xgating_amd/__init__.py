"""controllable_xgating_amd -- MI355X-native (gfx950) hot path of the gated-fusion caption decoder
(reference: vsislab/Controllable_XGating, caption_src/SAModel.py).  See DESIGN.md.

Environment note (the package never WRITES the environment; it reads two variables, both for tests / tools only: XG_LIBRARY --
another build of the same ABI, _native.py -- and XG_FORCE_DIST -- the collective path on one rank, train.py): a training process uses up to seven HIP
streams (the caller's, the library's two side streams, the optimizer's and the gradient all-reduce's side streams, RCCL's
own) and the ROCm runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); beyond that,
independent streams share a queue and serialise.  Launch data-parallel training with ``GPU_MAX_HW_QUEUES=8`` in the
environment (measured with the data-parallel path on one GPU: 7.44 ms per iteration with 4 queues, 6.18 with 8); it must
be set before the HIP runtime starts, which is why ``bench.py`` (a launcher) sets it and this package does not."""
from ._native import XgError, lib, LIB_PATH  # noqa: F401,E402
from .model import (SAModel, LanguageModelCriterion, ClassiferCriterion, RewardCriterion, make_opt)  # noqa: F401,E402

__all__ = ["SAModel", "LanguageModelCriterion", "ClassiferCriterion", "RewardCriterion", "make_opt", "XgError", "lib"]
