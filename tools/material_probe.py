"""Developer probe: wf_shade cost per shaded vertex for diffuse-only / conductor-only / mixed scenes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beifong_amd import capi, scenes, meshgen
from beifong_amd.scenedesc import SceneDesc, Transform4f as T

mesh = scenes.bus_mesh(200_000)
for name, gnd_kind, bus_kind in (("mixed", "d", "c"), ("all diffuse", "d", "d"), ("all conductor", "c", "c")):
    sd = SceneDesc()
    scenes._radar_frontend(sd)
    def mat(kind):
        return sd.add_diffuse(0.5, twosided=True) if kind == "d" else sd.add_roughconductor(alpha=0.1, twosided=True, specular_reflectance=1.0)
    sd.add_rectangle(T.translate([0, 0, 0]) * T.scale([20, 20, 1]), mat(gnd_kind))
    sd.add_mesh(mesh[0], mesh[1], mat(bus_kind), normals=mesh[2])
    sd.finalize()
    lp = capi.make_launch(capi.BF_MODE_RANGE, 1 << 24, seed=1, bins=256, bin_width=0.1, color_mode=capi.BF_COLOR_RGB)
    g = capi.Scene(sd)
    g.render(lp)
    h, _, st = g.render(lp)
    print(f"{name:14s} shade {st.shade_ms:6.2f} ms  trace {st.trace_ms:6.2f}  tail {st.tail_ms:6.2f}  bounces {st.n_bounces:10d}  "
          f"shade ns/vertex {st.shade_ms * 1e6 / st.n_bounces:6.3f}  rays {st.n_rays_closest + st.n_rays_shadow}", flush=True)
