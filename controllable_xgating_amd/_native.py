"""ctypes binding of libxgate_hip.so (the C ABI declared in include/xgate.h).

The library is built in-tree by ``__graft_entry__.build()``.  There is NO fallback: if the
shared object is missing or an entry point fails, the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# XG_LIBRARY: load another build of the same ABI instead (tests / tools: lib/libxgate_hip_diag.so, the -DXG_DIAG build
# that reads the diagnosis switches of DESIGN.md 6.2 from the environment; the product library reads none)
LIB_PATH = os.environ.get("XG_LIBRARY") or os.path.join(_HERE, "lib", "libxgate_hip.so")
LIB_DIAG_PATH = os.path.join(_HERE, "lib", "libxgate_hip_diag.so")

XG_VERSION = 206                      # include/xgate.h
XG_ROLLOUT_GREEDY, XG_ROLLOUT_SAMPLE, XG_ROLLOUT_REPLAY = 0, 1, 2


class XgDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "K", "R", "A", "E", "V", "C", "H", "F1", "F2", "T")]


class XgBnState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("rgb_mean", "rgb_var", "opfl_mean", "opfl_var")]


class XgBatch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("feats_rgb", "feats_opfl", "feat_mask", "pos_feats", "seq", "seq_mask")]


class XgRun(C.Structure):
    _fields_ = [("train", C.c_int32), ("drop_p", C.c_float), ("seed", C.c_uint32), ("save", C.c_int32),
                ("bn_momentum", C.c_float), ("bn_eps", C.c_float), ("gemm_mode", C.c_int32), ("packed_dtype", C.c_int32),
                ("packed", C.c_void_p), ("aux", C.c_void_p), ("grad_event", C.c_void_p), ("grad_event_head", C.c_void_p),
                ("prof_event0", C.c_void_p), ("prof_event1", C.c_void_p)]


class XgError(RuntimeError):
    pass


_lib = None
PARAM_NAMES = None
XgParams = None


def lib():
    """Load the HIP library once; raise loudly if it is not there."""
    global _lib, PARAM_NAMES, XgParams
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise XgError("libxgate_hip.so not found at %s -- build it with `python __graft_entry__.py` "
                      "(hipcc --offload-arch=gfx950); there is no CPU / PyTorch fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    # a stale build (older ABI) may lack entry points this binding declares: say so, instead of an AttributeError from ctypes
    need = ("xg_version", "xg_abi_check", "xg_strerror", "xg_param_count", "xg_param_name", "xg_param_numel", "xg_workspace_bytes",
            "xg_workspace_bytes_mode", "xg_packed_bytes")
    missing = [n for n in need if not hasattr(L, n)]
    if missing:
        raise XgError("%s lacks %s: a stale build of another ABI version -- rebuild it with `python __graft_entry__.py --force`"
                      % (LIB_PATH, ", ".join(missing)))
    L.xg_version.restype = C.c_int
    L.xg_strerror.restype = C.c_char_p
    L.xg_strerror.argtypes = [C.c_int]
    L.xg_param_count.restype = C.c_int
    L.xg_param_name.restype = C.c_char_p
    L.xg_param_name.argtypes = [C.c_int]
    L.xg_param_numel.restype = C.c_int
    L.xg_param_numel.argtypes = [C.POINTER(XgDims), C.c_int, C.POINTER(C.c_int64)]
    L.xg_workspace_bytes.restype = C.c_size_t
    L.xg_workspace_bytes.argtypes = [C.POINTER(XgDims)]
    L.xg_workspace_bytes_mode.restype = C.c_size_t
    L.xg_workspace_bytes_mode.argtypes = [C.POINTER(XgDims), C.c_int]
    L.xg_packed_bytes.restype = C.c_size_t
    L.xg_packed_bytes.argtypes = [C.POINTER(XgDims), C.c_int]
    n = L.xg_param_count()
    PARAM_NAMES = [L.xg_param_name(i).decode() for i in range(n)]

    class _XgParams(C.Structure):
        _fields_ = [("p%d" % i, C.c_void_p) for i in range(n)]

    XgParams = _XgParams
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    PD, PP, PB, PX, PR = C.POINTER(XgDims), C.POINTER(_XgParams), C.POINTER(XgBnState), C.POINTER(XgBatch), C.POINTER(XgRun)
    sigs = {
        "xg_gemm": [vp, i32, i32, i32, i32, i32, vp, i32, vp, i32, vp, i32, vp, i32, i32],
        "xg_gemm_mode": [vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, i32, vp, i32, vp, i32, i32],
        "xg_gemm_bf16_operands": [vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, vp, i32, vp, i32, i32],
        "xg_cvt_bf16": [vp, vp, vp, i64],
        "xg_encoder_fwd": [vp, PD, PP, PB, PX, PR, vp, C.c_size_t, vp],
        "xg_encoder_bwd": [vp, PD, PP, PP, PX, PR, vp, C.c_size_t, vp],
        "xg_init_hidden": [vp, PD, PP, vp, vp, vp, C.c_size_t, vp],
        "xg_vproj": [vp, PD, PP, vp, vp, PR],
        "xg_step_fwd": [vp, PD, PP, vp, vp, vp, vp, vp, PR, i32, vp, C.c_size_t, vp, vp, vp],
        "xg_step_bwd": [vp, PD, PP, PP, vp, vp, vp, vp, vp, PR, i32, vp, C.c_size_t, vp, vp, vp, vp, vp, vp],
        "xg_forward_xe": [vp, PD, PP, PB, PX, PR, vp, C.c_size_t, vp, vp],
        "xg_backward_xe": [vp, PD, PP, PP, PX, PR, vp, C.c_size_t, vp, vp],
        "xg_forward_ss": [vp, PD, PP, PB, PX, PR, f32, vp, vp, vp, C.c_size_t, vp, vp],
        "xg_backward_ss": [vp, PD, PP, PP, PX, PR, vp, C.c_size_t, vp, vp],
        "xg_xe_loss_fwd": [vp, PD, PP, PB, PX, vp, vp, f32, PR, vp, C.c_size_t, vp],
        "xg_xe_loss_bwd": [vp, PD, PP, PP, PX, vp, vp, f32, vp, PR, vp, C.c_size_t],
        "xg_rollout": [vp, PD, PP, PB, PX, PR, i32, vp, vp, f32, vp, C.c_size_t, vp, vp, vp],
        "xg_rollout_bwd": [vp, PD, PP, PP, PX, PR, vp, C.c_size_t, vp],
        "xg_aux_create": [C.POINTER(C.c_void_p)],
        "xg_aux_destroy": [vp],
        "xg_rollout_pair": [vp, PD, PP, PB, PX, PR, i32, vp, f32, vp, C.c_size_t, vp, vp, vp],
        "xg_rollout_compact": [vp, PD, vp, C.c_size_t, PD, vp, C.c_size_t],
        "xg_rollout_pair_compact": [vp, PD, PP, PB, PX, PR, i32, vp, f32, vp, C.c_size_t, PD, vp, C.c_size_t, vp, vp, vp],
        "xg_rollout_pair_videos": [vp, PD, PP, PB, PX, PR, vp, f32, vp, C.c_size_t, PD, vp, C.c_size_t, i32, vp, vp, vp],
        "xg_nll_fwd": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
        "xg_nll_bwd": [vp, vp, vp, vp, i32, i32, i32, i32, vp, f32, vp, vp],
        "xg_clip_adam": [vp, i64, vp, vp, vp, vp, f32, f32, f32, f32, f32, i32, f32],
        "xg_clip_adam_zero": [vp, i64, vp, vp, vp, vp, f32, f32, f32, f32, f32, i32, f32],
        "xg_workspace_init": [vp, vp, C.c_size_t],
        "xg_adam_tick": [vp, vp, f32, f32],
        "xg_clip_adam_dev": [vp, i64, vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, i32],
        "xg_pack_weights": [vp, PD, PP, vp, C.c_size_t, i32, i32],
        "xg_pack_weights_part": [vp, PD, PP, vp, C.c_size_t, i32, i32, i32],
        "xg_reward_fwd": [vp, vp, i32, vp, i32, vp, i32, i32, vp, i32, i32, vp],
        "xg_reward_bwd": [vp, vp, i32, vp, i32, i32, vp, i32, i32, vp, vp, vp, i32],
    }
    missing = [n for n in sigs if not hasattr(L, n)]
    if missing:
        raise XgError("%s (xg_version %d) lacks %s: rebuild it with `python __graft_entry__.py --force`"
                      % (LIB_PATH, L.xg_version(), ", ".join(missing)))
    for name, args in sigs.items():
        fn = getattr(L, name)
        fn.restype = C.c_int
        fn.argtypes = args
    L.xg_abi_check.restype = C.c_int
    L.xg_abi_check.argtypes = [C.c_int] + [C.c_size_t] * 5
    if L.xg_abi_check(XG_VERSION, C.sizeof(XgDims), C.sizeof(_XgParams), C.sizeof(XgBnState), C.sizeof(XgBatch), C.sizeof(XgRun)):
        raise XgError("%s has ABI version %d / other struct layouts than this binding (XG_VERSION %d): rebuild it with "
                      "`python __graft_entry__.py --force`" % (LIB_PATH, L.xg_version(), XG_VERSION))
    _lib = L
    return L


def check(rc: int, what: str):
    if rc != 0:
        raise XgError("%s failed: %s (code %d)" % (what, lib().xg_strerror(rc).decode(), rc))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def make_params_struct(tensors_by_name):
    L = lib()  # noqa: F841
    s = XgParams()
    for i, name in enumerate(PARAM_NAMES):
        setattr(s, "p%d" % i, tensors_by_name[name].data_ptr())
    return s
