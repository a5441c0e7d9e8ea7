"""avian_amd — MI355X-native physics step for Avian's 3D hot path.

The product is ``csrc/libavian_mi355x.so`` (hand-written HIP kernels for gfx950 + C++ host behind the C ABI
of ``include/avian_mi355x.h``).  This package is only the Python glue used by the tests and ``bench.py``:
a ctypes binding (:mod:`avian_amd._ffi`) and synthetic scene builders (:mod:`avian_amd.scenes`).

There is NO CPU fallback: :func:`load_library` raises if the HIP library has not been built, and creating a
world raises ``AVN_ERR_NO_DEVICE`` when no GPU is visible.
"""
from __future__ import annotations

import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AVN_LIB_PATH") or os.path.join(_HERE, "csrc", "libavian_mi355x.so")   # (AVN_LIB_PATH: A/B builds of the same library)
_lib = None


def load_library():
    """Load ``libavian_mi355x.so`` (once).  torch is imported first so that the process uses a single HIP
    runtime (torch bundles its own ``libamdhip64.so`` with the same SONAME)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). avian_amd has no CPU fallback.")
        import torch  # noqa: F401  (side effect: loads the HIP runtime the extension must share)
        from . import _ffi
        _lib = _ffi.Library(LIB_PATH, "avn_")
    return _lib


def create_world(cfg=None, **kw):
    """Create a device world (``avn_world_create``)."""
    from . import _ffi
    lib = load_library()
    if cfg is None:
        cfg = _ffi.default_config(**kw)
    return _ffi.World(lib, cfg)
