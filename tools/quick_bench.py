"""Developer timing helper: wall/kernel time of bf_render for the C2 scene."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import beifong_amd
beifong_amd.configure_runtime()
from beifong_amd import capi, scenes

n_paths = int(os.environ.get("PATHS", 1 << 24))
scene_name = os.environ.get("SCENE", "bus")
if scene_name == "bus":
    sd, lp = scenes.bus_radar(n_tris=200_000, n_paths=n_paths)
elif scene_name == "car":
    sd, lp = scenes.car_radar(n_tris=1_000_000, n_paths=n_paths)
else:
    sd, lp = scenes.multi_mesh_radar(n_paths=n_paths)
if os.environ.get('MAXDEPTH'):
    lp.max_depth = int(os.environ['MAXDEPTH'])
g = capi.Scene(sd)
for name, flags in (("wavefront", 0), ("megakernel", capi.BF_FLAG_MEGAKERNEL)):
    if os.environ.get("ONLY") and os.environ["ONLY"] != name:
        continue
    lp.flags = flags
    best = 1e9
    for i in range(4):
        t = time.time(); h, _, st = g.render(lp); dt = time.time() - t
        best = min(best, st.kernel_ms)
    rays = st.n_rays_closest + st.n_rays_shadow
    if os.environ.get("SPLIT"):
        print(f"           shade {st.shade_ms:6.2f}  trace {st.trace_ms:6.2f}  tail {st.tail_ms:6.2f}  iters {st.n_bounce_iters}")
    print(f"{name:10s} kernel {best:8.2f} ms  wall {dt*1e3:8.2f} ms  rays {rays}  {rays/best/1e3:8.1f} Mrays/s  {n_paths/best/1e3:8.1f} Mpaths/s", flush=True)
