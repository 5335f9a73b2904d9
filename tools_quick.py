import time, numpy as np
from beifong_amd import capi, scenes
t=time.time(); sd, lp = scenes.bus_radar(n_tris=200_000, n_paths=1<<22); print("scene build", time.time()-t)
t=time.time(); g = capi.Scene(sd); print("scene create", time.time()-t, g.info().n_bvh_nodes)
for flags in (0, capi.BF_FLAG_STATS):
    lp.flags = flags
    for i in range(3):
        t=time.time(); h, _, st = g.render(lp); dt=time.time()-t
        rays = st.n_rays_closest+st.n_rays_shadow
        print(f"flags={flags} wall {dt*1e3:.1f} ms kernel {st.kernel_ms:.2f} ms rays {rays} Mrays/s {rays/st.kernel_ms/1e3:.1f} nodes/ray {st.n_nodes_visited/max(rays,1):.1f} tris/ray {st.n_tris_tested/max(rays,1):.2f}")
