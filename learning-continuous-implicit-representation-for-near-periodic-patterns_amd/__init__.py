"""npp_amd -- MI355X-native NPP-Net optimisation path (host side).

Python here is plumbing (device memory, streams, torch.distributed); every operator of
the path runs in ``libnpp_hip.so`` (hand-written HIP for gfx950) through the C ABI declared
in ``include/npp_hip.h``.  There is no CPU fallback: if the library is missing or a symbol
is absent, ``lib()`` raises.
"""
from ._lib import lib, NppError, EmbedCfg, LIB_PATH, LIB_PATHS, FUSED_WIDTHS, SYMBOLS  # noqa: F401

__all__ = ["lib", "NppError", "EmbedCfg", "LIB_PATH", "SYMBOLS"]
