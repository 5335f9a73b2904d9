import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from beifong_amd import capi, scenes, meshgen
def info(sd):
    g = capi.Scene(sd); i = g.info(); return i.bvh_depth, i.bvh_stack_need, i.n_triangles
for n,seed in [(2000,3),(50000,4),(400000,5)]:
    v,f = meshgen.triangle_soup(n, seed=seed)
    print("soup", n, info(scenes.single_mesh(v,f)))
sd,_ = scenes.bus_radar(n_tris=200000, n_paths=1); print("bus", info(sd))
sd,_ = scenes.car_radar(n_tris=1000000, n_paths=1); print("car", info(sd))
# worst case for the stack: long thin overlapping slivers
v, f = meshgen.triangle_soup(200000, seed=9, extent=1.0, size=1.5)
print("big-soup", info(scenes.single_mesh(v, f)))
