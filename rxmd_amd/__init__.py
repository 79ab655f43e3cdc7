"""rxmd_amd -- MI355X-native ReaxFF force + charge-equilibration engine (the per-step hot path of
USCCACS/RXMD) behind a C ABI.  Hand-written HIP kernels for gfx950; no CPU fallback."""
from ._lib import load as load_library, SO_PATH  # noqa: F401
from .engine import RxmdEngine, RxmdError, PE_NAMES  # noqa: F401
from . import system  # noqa: F401
