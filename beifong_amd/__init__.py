"""beifong_amd — MI355X-native engine for beifong's transient-radar hot path.

The product is csrc/libbeifong_hip.so (hand-written HIP for gfx950 behind the
C ABI in include/beifong_hip.h).  This package is the thin host-side plumbing:
ctypes bindings (capi), the flat scene description builder (scenedesc) and the
Mitsuba-shaped front end.
"""
__version__ = "0.1.0"
