"""Loader of the HIP engine library (librmp2_hip.so) -- the ONLY compute backend.

There is deliberately no CPU / PyTorch fallback: if the library is missing, was built for a
different ABI, or no HIP device is usable, every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

from . import descriptor as D

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RMP2_LIB", os.path.join(_PKG, "librmp2_hip.so"))  # RMP2_LIB: diagnostic builds only

_lib = None


class Rmp2Error(RuntimeError):
    """code: the library's RMP2_ERR_* return value (include/rmp2.h) when the error came out of a C-ABI call, else None."""

    def __init__(self, message, code=None):
        super().__init__(message)
        self.code = code


ERR_INVALID_ARGUMENT, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_HIP, ERR_ABI_MISMATCH = -1, -2, -3, -4, -5


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Rmp2Error(
            f"{LIB_PATH} not found: the HIP engine is not built.  Run "
            "`python -c \"import __graft_entry__ as g; g.build()\"` at the repo root (needs hipcc). "
            "There is no CPU fallback.")
    # PyTorch first: the process then has ONE HIP runtime, PyTorch's (its bundled libamdhip64), which this library's HIP calls
    # resolve to as well.  Loaded the other way round -- this library before torch -- the process ends up with the system
    # runtime AND PyTorch's, and hipGetDeviceCount of the first answers "no device" (seen with build() followed by smoke() in
    # one process).  PyTorch is the plumbing for device memory and streams everywhere above this loader anyway.
    import torch  # noqa: F401
    l = C.CDLL(LIB_PATH)
    l.rmp2_abi_version.restype = C.c_int
    l.rmp2_sizeof_desc.restype = C.c_size_t
    l.rmp2_sizeof_obstacles.restype = C.c_size_t
    l.rmp2_last_error.restype = C.c_char_p
    l.rmp2_last_error.argtypes = [C.c_void_p]
    l.rmp2_last_kernel.restype = C.c_char_p
    l.rmp2_last_kernel.argtypes = [C.c_void_p]
    l.rmp2_validate.argtypes = [C.POINTER(D.Desc)]
    l.rmp2_fence_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    l.rmp2_fence_record.argtypes = [C.c_void_p, C.c_void_p]
    l.rmp2_fence_wait.argtypes = [C.c_void_p, C.c_void_p]
    l.rmp2_fence_destroy.argtypes = [C.c_void_p]
    l.rmp2_set_step_fence.argtypes = [C.c_void_p, C.c_void_p]
    l.rmp2_leaf_evaluate.argtypes = [C.c_int, C.POINTER(D.Leaf), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    l.rmp2_create.argtypes = [C.POINTER(D.Desc), C.c_int, C.POINTER(C.c_void_p)]
    l.rmp2_destroy.argtypes = [C.c_void_p]
    l.rmp2_reserve.argtypes = [C.c_void_p, C.c_int32]
    l.rmp2_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(D.Obstacles),
                            C.POINTER(D.Outputs), C.c_int32, C.c_void_p]
    _step_args = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(D.Obstacles), C.POINTER(D.Outputs), C.c_int32]
    l.rmp2_step_pair.argtypes = _step_args + _step_args + [C.c_void_p]
    l.rmp2_rollout.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(D.Obstacles),
                               C.POINTER(D.RolloutCfg), C.POINTER(D.Outputs), C.c_int32, C.c_void_p]
    l.rmp2_forward_kinematics.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    l.rmp2_closest_points.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(D.Obstacles), C.c_void_p, C.c_void_p,
                                      C.c_int32, C.c_void_p]
    l.rmp2_closest_points_links.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(D.Obstacles), C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int32, C.c_void_p]
    l.rmp2_differentiate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    l.rmp2_differentiate_euler.argtypes = l.rmp2_differentiate.argtypes
    l.rmp2_exchange_unique_id.argtypes = [C.c_char_p, C.c_void_p]
    l.rmp2_exchange_create.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    l.rmp2_exchange_destroy.argtypes = [C.c_void_p]
    l.rmp2_exchange_last_error.restype = C.c_char_p
    l.rmp2_exchange_last_error.argtypes = [C.c_void_p]
    l.rmp2_exchange_pending.argtypes = [C.c_void_p]
    l.rmp2_exchange_nranks.argtypes = [C.c_void_p]
    l.rmp2_exchange_set_depth.argtypes = [C.c_void_p, C.c_int32]
    l.rmp2_exchange_set_peer_wait.argtypes = [C.c_void_p, C.c_int32]
    l.rmp2_exchange_start.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    l.rmp2_exchange_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                     C.c_int32, C.POINTER(D.Outputs), C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]
    if l.rmp2_abi_version() != D.ABI_VERSION:
        raise Rmp2Error(f"ABI mismatch: library {l.rmp2_abi_version()} vs bindings {D.ABI_VERSION}")
    if l.rmp2_sizeof_desc() != C.sizeof(D.Desc) or l.rmp2_sizeof_obstacles() != C.sizeof(D.Obstacles):
        raise Rmp2Error("struct layout mismatch between include/rmp2.h and descriptor.py")
    _lib = l
    return l


def check(rc: int, handle=None):
    if rc != 0:
        msg = lib().rmp2_last_error(handle)
        raise Rmp2Error(f"rmp2 error {rc}: {msg.decode() if msg else '?'}", code=rc)
