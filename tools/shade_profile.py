"""Developer tool: lane utilisation of wf_shade by section.  Needs the instrumented build
(make -C beifong_amd/csrc variant VARIANT=shprof EXTRA=-DBF_SHADE_PROF) and BF_HIP_LIB=beifong_amd/csrc/libbeifong_hip_shprof.so.
Renders the C2 bench step (200 k bus, PATHS paths) and prints, per section, wave entries, lanes per entry and the wave
cycles between the section stamps."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from beifong_amd import capi, scenes

n_paths = int(os.environ.get("PATHS", 1 << 24))
scene_name = os.environ.get("SCENE", "bus")
if scene_name == "bus":
    sd, lp = scenes.bus_radar(n_tris=200_000, n_paths=n_paths)
elif scene_name == "c5":        # one C5-like pulse batch: I/Q receive on the bus, the pool a quarter of the launch (slots regenerate)
    lam = 8.6e6
    sd, lp = scenes.bus_receive(n_tris=200_000, n_paths=n_paths, t_bins=1024, dr=0.03, seed=4, lambda_band_nm=(lam * 0.999, lam * 1.001))
    lp.mode = capi.BF_MODE_RECEIVE_IQ
elif scene_name == "car":
    sd, lp = scenes.car_radar(n_tris=1_000_000, n_paths=n_paths)
else:
    sd, lp = scenes.multi_mesh_radar(n_paths=n_paths)
lib = capi.load_library()
lib.bfdbg_shade_lane_profile.argtypes = [C.c_void_p, C.c_int]
g = capi.Scene(sd, lib)
g.render(lp)
lib.bfdbg_shade_lane_profile(None, 1)
h, _, st = g.render(lp)
buf = np.zeros(64, dtype=np.uint64)
assert lib.bfdbg_shade_lane_profile(buf.ctypes.data_as(C.c_void_p), 0) == 64
w, l = buf[:32].astype(np.float64), buf[32:].astype(np.float64)
names = {0: "batch visit (has)", 1: "  load: term-pending -> film_put", 2: "vertex round 0", 3: "vertex round 1", 4: "vertex round 2",
         5: "  film_put after vertex", 6: "generate_path", 7: "presolve shadow", 8: "presolve closest", 9: "chain: term-pending film_put",
         10: "  make_si", 11: "  emitter pdf (depth > 0)", 12: "  emitter head", 13: "  NEE + BSDF (survivors)", 14: "  NEE bsdf eval",
         15: "  roughconductor lanes", 16: "  diffuse lanes", 17: "chain after round 0", 18: "chain after round 1", 19: "chain after round 2",
         20: "store state", 21: "store shadow ray"}
print(f"{scene_name} {n_paths} paths: kernel {st.kernel_ms:.2f} ms shade {st.shade_ms:.2f} trace {st.trace_ms:.2f} tail {st.tail_ms:.2f}"
      f" iters {st.n_bounce_iters} bounces {st.n_bounces} (instrumented build: times are not the product's)")
print(f"{'section':38s} {'wave entries':>12s} {'lanes':>12s} {'lanes/entry':>11s}")
for k in sorted(names):
    if w[k]:
        print(f"{names[k]:38s} {int(w[k]):12d} {int(l[k]):12d} {l[k] / w[k]:11.1f}")
t = w[24:32]
tn = ["cursor + gather", "state / hit load", "film_put after a vertex", "generate_path", "presolve", "chain bookkeeping", "store + masks", "shade_vertex"]
print("wave cycles between stamps (s_memtime ticks), share of the total:")
for k in range(8):
    print(f"  {tn[k]:24s} {t[k] / t[:8].sum():6.3f}")
v = l[24:28]
if v.sum():
    print("inside vertex + film_put (largest lane of a wave; the rest of that share is film_put and lanes that end early):")
    for k, nm in enumerate(["surface interaction (+ wait for the triangle / material)", "head (emitter, roulette)", "next-event estimation", "BSDF sampling + spawn"]):
        print(f"  {nm:58s} {v[k] / t[:8].sum():6.3f}")
