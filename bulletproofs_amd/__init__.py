"""MI355X-native MSM + Bulletproofs-IPA folding (hot path of alex-ozdemir/bulletproofs).

The product is the HIP library behind include/bpgpu.h; this package is the thin host
side that loads it.  Nothing here computes on the CPU.
"""
from . import ffi  # noqa: F401

__all__ = ["ffi"]
