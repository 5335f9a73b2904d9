"""beifong_amd — MI355X-native engine for beifong's transient-radar hot path.

The product is csrc/libbeifong_hip.so (hand-written HIP for gfx950 behind the
C ABI in include/beifong_hip.h).  This package is the thin host-side plumbing:
ctypes bindings (capi), the flat scene description builder (scenedesc) and the
Mitsuba-shaped front end.
"""
__version__ = "0.1.0"

import os as _os


def configure_runtime(hw_queues=16):
    """Ask the HIP runtime for `hw_queues` hardware queues per process (GPU_MAX_HW_QUEUES, default 4).

    The runtime multiplexes a process's streams onto that many queues: with the default, at most three renders overlap
    and a fourth stream is slower than three; sweeps and pipelined renders keep more in flight.  The
    variable is read ONCE, when the runtime starts, so call this before the process's first HIP call (`import torch` alone
    does not start it); an embedding host that has its own opinion sets the variable itself or never calls this.
    Returns the value now in the environment.  Explicit on purpose: importing the package changes nothing global
    (bench.py, PulseSweeper and the tools call it)."""
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", str(int(hw_queues)))
    return _os.environ["GPU_MAX_HW_QUEUES"]
