"""yolo2_light_amd -- MI355X (gfx950) native hot path for AlexeyAB/yolo2_light.

The product is libyolo2hip.so (hand-written HIP kernels behind the C-ABI in
include/yolo2_hip.h); this package is the thin host-side mirror used by the
tests and bench.py.  Importing it requires the built library: there is no
CPU fallback.
"""
from ._lib import LIB_PATH, YoloHipError, lib          # noqa: F401  (fails loudly if the .so is missing)
from .network import Network                            # noqa: F401
from . import zoo, weights                              # noqa: F401

__all__ = ["Network", "YoloHipError", "zoo", "weights", "LIB_PATH"]
