"""ctypes binding of libsind_hip.so (the product's C ABI, include/sind_hip.h).

There is no CPU fallback: if the HIP library is missing or fails to load this module raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libsind_hip.so")


class SindError(RuntimeError):
    pass


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise SindError(f"{SO_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950)")
        # PyTorch's ROCm wheel bundles its own libamdhip64 / libhsa-runtime64.  When this process also uses torch, let torch
        # bring its HIP runtime up FIRST: initialising the system runtime (which libsind_hip links) before torch's copy has been
        # seen to leave torch with "No HIP GPUs are available".  Plumbing only; the library itself never calls into torch.
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
        _lib = C.CDLL(SO_PATH)
        _lib.sind_last_error.restype = C.c_char_p
        _lib.sind_pipe_state_bytes.restype = C.c_size_t
    return _lib


def check(rc: int, what: str = ""):
    if rc < 0:
        raise SindError(f"{what} failed ({rc}): {lib().sind_last_error().decode(errors='replace')}")
    return rc


def ptr(a):
    """numpy array -> void*, None -> NULL, int -> device pointer."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)
