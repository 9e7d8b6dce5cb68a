"""lamp_amd — MI355X-native loop-closure verification engine for LAMP.

The product is the C-ABI library lamp_amd/liblamp_gicp.so (HIP, gfx950).  This package only
loads it (lamp_amd.capi) and mirrors the reference's LoopComputation interface on top of it
(lamp_amd.loop_computation).  There is no CPU fallback: without the built library or without
a gfx950 device every call fails loudly.
"""
from . import capi  # noqa: F401
