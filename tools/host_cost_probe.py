"""Host-side cost of issuing rolling calls: time the host spends per bf_render_device(BF_FLAG_ROLLING) against the GPU time of the sequence."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beifong_amd
beifong_amd.configure_runtime()
from beifong_amd import capi, scenes
for name, (sd, lp) in (("c3", scenes.car_radar(n_tris=1_000_000, n_paths=1 << 20, bins=1024, dr=0.03)),
                       ("tiny 2^16 paths", scenes.bus_radar(n_tris=20000, n_paths=1 << 16, bins=256, dr=0.1))):
    g = capi.Scene(sd)
    n = g.channels(lp)
    K = 40
    h = torch.zeros((K, n), device="cuda")
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            l = capi.make_launch(lp.mode, lp.n_paths, seed=100 + k, bins=lp.bins, bin_width=lp.bin_width, flags=capi.BF_FLAG_ROLLING)
            g.render_device(l, h[k].data_ptr())
        t1 = time.perf_counter()
        g.flush(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name}: host time per rolling call {(t1 - t0) / K * 1e6:.1f} us; whole sequence {(t2 - t0) / K * 1e6:.1f} us per render")
