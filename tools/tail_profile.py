"""Developer tool: where the tail kernel's time goes.  Needs the instrumented build (make -C beifong_amd/csrc prof) and
BF_HIP_LIB=beifong_amd/csrc/libbeifong_hip_prof.so.  Renders one isolated render of SCENE (bus / car / multi) with PATHS
paths and prints, for the waves with the most loop iterations, the cycles spent in each phase of the tail loop."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from beifong_amd import capi, scenes

n_paths = int(os.environ.get("PATHS", 1 << 20))
scene_name = os.environ.get("SCENE", "bus")
if scene_name == "bus":
    sd, lp = scenes.bus_radar(n_tris=200_000, n_paths=n_paths)
elif scene_name == "car":
    sd, lp = scenes.car_radar(n_tris=1_000_000, n_paths=n_paths)
else:
    sd, lp = scenes.multi_mesh_radar(n_paths=n_paths)
lib = capi.load_library()
g = capi.Scene(sd, lib)
for i in range(3):
    lib.bfdbg_tail_profile_clear()
    h, _, st = g.render(lp)
print(f"{scene_name} {n_paths} paths: kernel {st.kernel_ms:.2f} ms shade {st.shade_ms:.2f} trace {st.trace_ms:.2f} tail {st.tail_ms:.2f} iters {st.n_bounce_iters}")
n = 8192
buf = np.zeros((n, 24), dtype=np.uint64)
lib.bfdbg_tail_profile.argtypes = [C.c_void_p, C.c_int]
got = lib.bfdbg_tail_profile(buf.ctypes.data_as(C.c_void_p), n)
assert got == n
live = buf[buf[:, 0] > 0]
print(f"waves {len(live)}; iterations: mean {live[:, 0].mean():.1f} max {live[:, 0].max()}; total cycles (s_memtime, 100 MHz ticks?) max {live[:, 6].max()}")
order = np.argsort(-live[:, 6].astype(np.int64))[:8]
print(" iters   regen    trav    film   shade  quad_it   total  | per-iter: regen trav film shade")
for k in order:
    it, rg, tr, fm, sh, qd, tot, nn = [int(x) for x in live[k][:8]]
    rpass, rsteps, rrect, rmem, rcmp = [int(x) for x in live[k][8:13]]
    ssi, shead, snee = [int(x) for x in live[k][13:16]]
    sbsdf = nn
    print(f"{it:6d} {rg:7d} {tr:7d} {fm:7d} {sh:7d} {qd:7d} {tot:8d} | {rg / it:7.1f} {tr / it:7.1f} {fm / it:7.1f} {sh / it:7.1f}"
          f" | row passes {rpass} steps/pass {rsteps / max(rpass, 1):.1f} rect/pass {rrect / max(rpass, 1):.0f} mem/step {rmem / max(rsteps, 1):.0f} cmp/step {rcmp / max(rsteps, 1):.0f}"
          f" | shade per iter (lane 0's completed vertices): si {ssi / it:.0f} head {shead / it:.0f} nee {snee / it:.0f} bsdf {sbsdf / it:.0f}")
di, dt, ds, dr = [live[:, k].astype(np.float64) for k in (16, 17, 18, 19)]
print(f"dense iterations (> 16 rays in the wave): {di.sum():.0f} over {len(live)} waves, mean per wave {di.mean():.1f}; cycles per dense iteration: "
      f"traversal {dt.sum() / max(di.sum(), 1):.0f}, shading {ds.sum() / max(di.sum(), 1):.0f}; rays per dense iteration {dr.sum() / max(di.sum(), 1):.1f}")
print(f"share of the waves' total cycles spent in dense iterations: traversal {dt.sum() / live[:, 6].sum():.2f}, shading {ds.sum() / live[:, 6].sum():.2f}; "
      f"all traversal {live[:, 2].sum() / live[:, 6].sum():.2f}, all shading {live[:, 4].sum() / live[:, 6].sum():.2f}")
tot = live[:, 6].astype(np.float64)
print(f"tail_ms {st.tail_ms:.3f} -> ticks per ms of the slowest wave: {tot.max() / st.tail_ms:.0f}")
