"""amaranth_twstft_amd — MI355X-native TWSTFT correlation post-processing (hot path only).

See DESIGN.md.  ``prn`` and ``synth`` are host utilities (numpy); ``correlator`` drives the
HIP library through its C ABI and has no CPU fallback.
"""
from . import prn, synth  # noqa: F401

__all__ = ["prn", "synth", "correlator"]
