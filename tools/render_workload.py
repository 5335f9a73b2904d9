"""Developer tool: a few isolated renders of a bench workload, to be run under rocprofv3 (--kernel-trace or --pmc passes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beifong_amd import capi, scenes

n_paths = int(os.environ.get("PATHS", 1 << 24))
reps = int(os.environ.get("REPS", 4))
sd, lp = scenes.bus_radar(n_tris=200_000, n_paths=n_paths)
g = capi.Scene(sd)
for i in range(reps):
    h, _, st = g.render(lp)
print(f"kernel {st.kernel_ms:.2f} ms shade {st.shade_ms:.2f} trace {st.trace_ms:.2f} tail {st.tail_ms:.2f}")
