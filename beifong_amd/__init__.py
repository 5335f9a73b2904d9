"""beifong_amd — MI355X-native engine for beifong's transient-radar hot path.

The product is csrc/libbeifong_hip.so (hand-written HIP for gfx950 behind the
C ABI in include/beifong_hip.h).  This package is the thin host-side plumbing:
ctypes bindings (capi), the flat scene description builder (scenedesc) and the
Mitsuba-shaped front end.
"""
__version__ = "0.1.0"

import os as _os

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): with the default, at
# most three renders overlap and a fourth stream is slower than three.  Sweeps and pipelined renders keep
# more in flight, so the package asks for 16 — if the variable is unset and this import comes before the process's first
# HIP call (the runtime reads it when it starts; `import torch` alone does not start it).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

