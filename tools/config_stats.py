"""V_n / V_t / B_ray and timings of BASELINE's configs on the shipped four-wide BVH (instrumented kernels, BF_FLAG_STATS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beifong_amd import capi, scenes

cfgs = [("C2", lambda: scenes.bus_radar(n_tris=200_000, n_paths=1 << 24)),
        ("C3", lambda: scenes.car_radar(n_tris=1_000_000, n_paths=1 << 20, bins=1024, dr=0.03)),
        ("C4 (one of 8 shards)", lambda: scenes.multi_mesh_radar(n_paths=(4096 << 10) // 8))]
for name, make in cfgs:
    sd, lp = make()
    g = capi.Scene(sd)
    info = g.info()
    lp.flags = capi.BF_FLAG_STATS
    _, _, st = g.render(lp)
    rays = st.n_rays_closest + st.n_rays_shadow
    vn, vt = st.n_nodes_visited / rays, st.n_tris_tested / rays
    b = vn * info.node_bytes + vt * info.tri_bytes + 48
    lp.flags = 0
    g.render(lp)
    best = min(g.render(lp)[2].kernel_ms for _ in range(3))
    print(f"{name}: tris {info.n_triangles} nodes4 {info.n_bvh_nodes} depth {info.bvh_depth} stack {info.bvh_stack_need} | paths {lp.n_paths} "
          f"rays {rays} | V_n {vn:.2f} V_t {vt:.2f} B_ray {b:.0f} B ceiling {8e12 / b / 1e9:.1f} Grays/s | isolated {best:.2f} ms "
          f"= {rays / best / 1e3:.0f} Mrays/s", flush=True)
