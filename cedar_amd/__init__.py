"""cedar_amd -- MI355X-native BoxMG V-cycle hot path (host-side Python mirror).

The product is the C-ABI shared library ``cedar_amd/lib/libcedar_amd.so`` built
from ``cedar_amd/csrc`` (hand-written HIP for gfx950).  This package is a thin
ctypes front-end over that ABI for tests and benchmarks; it never falls back to
a CPU implementation: importing :mod:`cedar_amd.capi` without the built library
raises.
"""
__version__ = "0.1"
