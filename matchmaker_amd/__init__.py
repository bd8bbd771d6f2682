"""matchmaker_amd — MI355X-native (gfx950) interaction scoring for matchmaker's re-ranking forward
pass: ColBERT MaxSim and TK / TKL kernel pooling as hand-written HIP kernels behind the reference's
own model interface.  See DESIGN.md / INTEGRATION.md."""
from ._lib import NativeError, LIB_PATH  # noqa: F401
from . import ops  # noqa: F401

__all__ = ["ops", "NativeError", "LIB_PATH"]
