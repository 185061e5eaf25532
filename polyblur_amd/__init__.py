"""polyblur_amd -- MI355X-native Polyblur blind deblurring (drop-in for teboli/polyblur's
``polyblur_deblurring`` / ``PolyblurDeblurring``; reference polyblur/__init__.py:1).

The compute path is hand-written HIP for gfx950 behind the C ABI in
``include/polyblur_hip.h`` (``libpolyblur_hip.so``), loaded lazily with ctypes on the
first call.  There is no CPU fallback: without the library or a GPU the calls raise.
"""
from .deblurring import polyblur_deblurring, polyblur_deblurring_uint8, PolyblurDeblurring  # noqa: F401

__all__ = ["polyblur_deblurring", "polyblur_deblurring_uint8", "PolyblurDeblurring"]
__version__ = "0.1.0"
