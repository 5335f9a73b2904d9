"""Developer probe: K renders of the C2 scene issued on 1 vs 2 HIP streams (one bf_scene per
stream).  Shows how much of the deep-path tail overlaps with the next render's head."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beifong_amd import capi, scenes

n_paths = int(os.environ.get("PATHS", 1 << 24))
K = int(os.environ.get("K", 8))
sd, lp = scenes.bus_radar(n_tris=200_000, n_paths=n_paths)
flags = capi.BF_FLAG_MEGAKERNEL if os.environ.get("MODE", "megakernel") == "megakernel" else 0
lp.flags = flags
if os.environ.get('MAXDEPTH'):
    lp.max_depth = int(os.environ['MAXDEPTH'])
lib = capi.load_library()
nch = lib.bf_launch_channels(lp)
for n_streams in [int(x) for x in os.environ.get("STREAMS", "1,2,3").split(",")]:
    sc = [capi.Scene(sd) for _ in range(n_streams)]
    st = [torch.cuda.Stream() for _ in range(n_streams)]
    hist = [torch.zeros(nch, device="cuda") for _ in range(K)]
    def run():
        for k in range(K):
            j = k % n_streams
            sc[j].render_device(lp, hist[k].data_ptr(), stream=st[j].cuda_stream)
    run(); torch.cuda.synchronize()
    for h in hist: h.zero_()
    torch.cuda.synchronize()
    t = time.time(); run(); torch.cuda.synchronize(); dt = time.time() - t
    same = all(torch.allclose(hist[0], h, rtol=1e-4, atol=1e-6) for h in hist)
    print(f"streams={n_streams}  {dt / K * 1e3:7.2f} ms/render  all hist equal: {same}", flush=True)
