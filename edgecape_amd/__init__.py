"""edgecape_amd — MI355X-native EdgeCape inference hot path.

Python face = the reference's registry / detector API; compute = libedgecape_hip.so (hand-written
gfx950 HIP kernels behind the C ABI of include/edgecape_hip.h).
"""
from .registry import (HEADS, POSENETS, POSITIONAL_ENCODING, TRANSFORMER, build_head, build_posenet,  # noqa: F401
                       build_positional_encoding, build_transformer)
from .config import Config  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):
    # torch / the HIP library are imported lazily so that `import edgecape_amd` stays cheap
    if name in ("EdgeCape", "load_checkpoint"):
        from . import detector
        return getattr(detector, name)
    if name == "HipEngine":
        from .engine import HipEngine
        return HipEngine
    raise AttributeError(name)
