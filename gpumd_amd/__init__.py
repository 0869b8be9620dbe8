"""gpumd_amd -- MI355X-native NEP force engine + velocity-Verlet path (drop-in for GPUMD's
src/force/nep*, src/force/neighbor*, src/integrate velocity-Verlet).

The product is the C-ABI library gpumd_amd/lib/libnepmi.so (hand-written HIP kernels for gfx950,
include/nepmi.h).  This package is the thin Python host mirror used by bench.py and the tests; it
uses torch only for device memory, streams and torch.distributed.  There is NO CPU fallback: if
the HIP library or a GPU is missing, construction fails loudly.
"""
import ctypes
import os

from . import _capi
from ._capi import NepmiError, NepmiInfo, NepmiStats  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libnepmi.so")
_lib = None


def load_library():
    """Load and bind gpumd_amd/lib/libnepmi.so (built by __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "gpumd_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        # torch ships its own libamdhip64.so.7; importing it first makes the HIP runtime a single
        # shared instance (same SONAME) so device pointers are interchangeable.
        import torch  # noqa: F401
        _lib = _capi.bind(ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL))
    return _lib


from .nep import NEP, Model  # noqa: E402,F401
